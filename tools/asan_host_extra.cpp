// What host_io.cpp needs from the device translation units when it is built alone (tools/asan_host.sh): the name hash,
// whose one definition is gci_common.h.
#include "../gci_amd/csrc/gci_common.h"
extern "C" uint64_t gci_name_hash(const uint8_t* name, uint32_t len)
{
    uint64_t acc = 0;
    for (uint32_t k = 0; k * 8 < len; k++) {
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) if (k * 8 + b < len) w |= (uint64_t)name[k * 8 + b] << (8 * b);
        acc += gci_hash_word(w, k);
    }
    return gci_hash_finish(acc, len);
}

// (hostio.py formats its error messages with it)
#include "../include/gci_hip.h"
extern "C" const char* gci_strerror(int s)
{
    switch (s) {
    case GCI_OK: return "ok";
    case GCI_E_INVALID: return "invalid argument";
    case GCI_E_ZERO_DIV: return "zero denominator (reference raises ZeroDivisionError, GCI.py:165/292)";
    case GCI_E_MALFORMED: return "malformed BAM record";
    case GCI_E_CAPACITY: return "output capacity exceeded";
    case GCI_E_NOMEM: return "out of memory";
    default: return "status";
    }
}
