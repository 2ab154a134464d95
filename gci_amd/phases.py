"""Where a run of the command line spends its time (number 3): wall-clock phases of the host and, inside the
ingestion, device time per stage from HIP events on the engine's stream (recorded without synchronising: the uploads, the
inflate and the record filter of consecutive runs overlap, and a device synchronisation per stage would undo exactly that).

Off by default -- the command line prints and writes what the reference does and nothing else.  `GCI_PHASES=<file.json>` (or
`phases.start()` from a harness such as bench.py) switches it on; `phases.report()` returns / writes the split."""
from __future__ import annotations

import contextlib
import json
import os
import time
from typing import Dict, List, Optional, Tuple

_ON = False
_WALL: List[Tuple[str, float]] = []
_GPU: List[Tuple[str, object, object]] = []
_NOTES: Dict[str, object] = {}
_T0 = 0.0
_TRACE = os.environ.get("GCI_PHASES_TRACE", "0") == "1"
_BASE = None


def process_age() -> float:
    """Seconds since this process was started (exec), from /proc: what of a run's wall time lies in front of / behind the phase log."""
    try:
        ticks = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[19])
        up = float(open("/proc/uptime").read().split()[0])
        return up - ticks / os.sysconf("SC_CLK_TCK")
    except Exception:                                      # noqa: BLE001
        return -1.0


def start() -> None:
    global _ON, _T0
    _ON = True
    _WALL.clear()
    _GPU.clear()
    _NOTES.clear()
    global _BASE
    _BASE = None
    _T0 = time.perf_counter()
    _NOTES["process_age_s_when_the_phase_clock_started"] = round(process_age(), 3)


def stop() -> None:
    global _ON
    _ON = False


def on() -> bool:
    return _ON


def env_start() -> Optional[str]:
    """Called by the command line: GCI_PHASES=<path> switches the log on; -> the path."""
    p = os.environ.get("GCI_PHASES")
    if p:
        start()
    return p or None


@contextlib.contextmanager
def wall(name: str):
    """A host wall-clock phase (nested phases are listed with their own names; they do not subtract from the outer one)."""
    if not _ON:
        yield
        return
    t = time.perf_counter()
    try:
        yield
    finally:
        _WALL.append((name, time.perf_counter() - t))


_PROVIDER = None


def set_provider(T) -> None:
    """The buffer / stream provider (gci_amd/hbm.py) whose events time the device stages: an Engine says which when it is made."""
    global _PROVIDER
    _PROVIDER = T


def _provider():
    if _PROVIDER is not None:
        return _PROVIDER
    from . import hbm
    return hbm.provider()


@contextlib.contextmanager
def gpu(name: str, stream=None):
    """Device time of what is enqueued inside, between two events on the current (= the engine's) stream."""
    if not _ON:
        yield
        return
    T = _provider()
    global _BASE
    if _TRACE and _BASE is None:                          # a common clock: an event right behind a synchronisation + the host's time then
        T.synchronize()
        e = T.Event(enable_timing=True)
        e.record(stream)
        e.synchronize()
        _BASE = (e, now())
    a, b = T.Event(enable_timing=True), T.Event(enable_timing=True)
    a.record(stream)
    try:
        yield
    finally:
        b.record(stream)
        _GPU.append((name, a, b))


def now() -> float:
    """Seconds since the log was started (0.0 when it is off)."""
    return time.perf_counter() - _T0 if _ON else 0.0


def trace(kind: str, *values) -> None:
    """One line of the per-run trace of the ingestion (GCI_PHASES_TRACE=1): kept in the notes as "trace"."""
    if _ON and _TRACE:
        _NOTES.setdefault("trace", []).append([kind] + [round(v, 4) if isinstance(v, float) else v for v in values])


def note(key: str, value) -> None:
    if _ON:
        _NOTES[key] = value


def add(key: str, value) -> None:
    if _ON:
        _NOTES[key] = _NOTES.get(key, 0) + value


def report(path: Optional[str] = None) -> dict:
    """{"wall_s": {phase: seconds (summed over its occurrences)}, "gpu_s": {stage: seconds}, "notes": {...}, "total_s"}."""
    out = {"total_s": time.perf_counter() - _T0, "wall_s": {}, "gpu_s": {}, "notes": dict(_NOTES)}
    out["notes"]["process_age_s_at_the_report"] = round(process_age(), 3)
    for name, s in _WALL:
        out["wall_s"][name] = out["wall_s"].get(name, 0.0) + s
    if _GPU:
        _provider().synchronize()
        # per stage also its FIRST occurrence, its longest one and their number: a first call that takes ten times the later ones (the
        # first inflate of a file on some boxes: DESIGN.md section 8) shows in every phase log, traced or not
        out["gpu_first_s"], out["gpu_max_s"], out["gpu_calls"] = {}, {}, {}
        for name, a, b in _GPU:
            dt = a.elapsed_time(b) * 1e-3
            out["gpu_s"][name] = out["gpu_s"].get(name, 0.0) + dt
            out["gpu_first_s"].setdefault(name, dt)
            out["gpu_max_s"][name] = max(out["gpu_max_s"].get(name, 0.0), dt)
            out["gpu_calls"][name] = out["gpu_calls"].get(name, 0) + 1
        if _TRACE and _BASE is not None:                  # every stage on the host's clock: [name, begin, end]
            e0, t0 = _BASE
            out["gpu_trace"] = [[name, round(t0 + e0.elapsed_time(a) * 1e-3, 4), round(t0 + e0.elapsed_time(b) * 1e-3, 4)] for name, a, b in _GPU]
    if path:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    return out
