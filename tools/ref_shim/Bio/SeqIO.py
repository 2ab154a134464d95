"""Stand-in for Bio.SeqIO.parse(path, 'fasta'): yields objects with .id and .seq, following
Biopython's FASTA conventions (id = title up to first whitespace; line ends stripped; blanks
and carriage returns removed from the sequence)."""


class _Rec:
    def __init__(self, rid, seq):
        self.id, self.seq = rid, seq


def parse(path, fmt):
    assert fmt == "fasta"
    rid, lines = None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if rid is not None:
                    yield _Rec(rid, "".join(lines).replace(" ", "").replace("\r", ""))
                title = line[1:].rstrip()
                parts = title.split(None, 1)
                rid, lines = (parts[0] if parts else ""), []
            elif rid is not None:
                lines.append(line.rstrip())
    if rid is not None:
        yield _Rec(rid, "".join(lines).replace(" ", "").replace("\r", ""))
