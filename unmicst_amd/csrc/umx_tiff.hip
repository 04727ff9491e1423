// TIFF strip / tile decoders of libumx (host code; used by unmicst_amd/tiffio.py through the C ABI).
#include <cstdint>
#include <cstring>

#include "../../include/umx.h"

extern "C" {

// ---- TIFF strip / tile decoders for the drivers' file reader (unmicst_amd/tiffio.py).  Host code: file decoding is not on
// the GPU path; it lives in the library so that the reader does not depend on tifffile / imagecodecs (absent here), which
// the reference uses at UnMicst1-5.py:794-797.  Both return the number of bytes written, or -1 on a malformed stream.
long long umx_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    // TIFF 6.0 LZW: codes packed MSB first, 9..12 bits, ClearCode 256, EndOfInformation 257, "early change" (the width
    // grows one code before the table fills a power of two), as written by libtiff, Bio-Formats and tifffile
    static thread_local uint16_t prefix[4096];
    static thread_local uint8_t suffix[4096];
    static thread_local uint16_t length[4096];
    if (!src || !dst) return -1;
    for (int i = 0; i < 256; ++i) { prefix[i] = 0; suffix[i] = (uint8_t)i; length[i] = 1; }
    int bits = 9, next = 258, prev = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t out = 0, ip = 0;
    for (;;) {
        while (nacc < bits && ip < n) { acc = (acc << 8) | src[ip++]; nacc += 8; }
        if (nacc < bits) break;                       // stream ended without EOI: accept what was decoded
        const int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1));
        nacc -= bits;
        if (code == 256) { bits = 9; next = 258; prev = -1; continue; }
        if (code == 257) break;
        if (prev < 0) {
            if (code > 255) return -1;
            if (out < cap) dst[out] = (uint8_t)code;
            ++out;
            prev = code;
            continue;
        }
        int entry;
        uint8_t first;
        if (code < next) {
            entry = code;
        } else if (code == next) {
            entry = prev;                             // KwKwK: the string of prev + its own first character
        } else {
            return -1;
        }
        // first character of `entry`'s string
        int e = entry;
        while (length[e] > 1) e = prefix[e];
        first = suffix[e];
        const size_t len = length[entry] + (code == next ? 1u : 0u);
        if (out + len <= cap) {
            size_t pos = out + length[entry];
            e = entry;
            while (true) {
                dst[--pos] = suffix[e];
                if (length[e] == 1) break;
                e = prefix[e];
            }
            if (code == next) dst[out + len - 1] = first;
        }
        out += len;
        if (next < 4096) {
            prefix[next] = (uint16_t)prev;
            suffix[next] = first;
            length[next] = (uint16_t)(length[prev] + 1);
            ++next;
            if (next >= (1 << bits) - 1 && bits < 12) ++bits;
        }
        prev = code;
    }
    return out <= cap ? (long long)out : -1;
}

long long umx_tiff_packbits_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (!src || !dst) return -1;
    size_t ip = 0, out = 0;
    while (ip < n) {
        const int8_t h = (int8_t)src[ip++];
        if (h >= 0) {                                  // h + 1 literal bytes
            const size_t k = (size_t)h + 1;
            if (ip + k > n || out + k > cap) return -1;
            memcpy(dst + out, src + ip, k);
            ip += k; out += k;
        } else if (h != -128) {                        // next byte repeated 1 - h times
            const size_t k = (size_t)(1 - h);
            if (ip >= n || out + k > cap) return -1;
            memset(dst + out, src[ip++], k);
            out += k;
        }
    }
    return (long long)out;
}

}  // extern "C"
