/* The reference's FRIES/math_utils.h for the MI355X build: same names and signatures, bit strings of at most 64 bits handled as
 * one word (reference: byte loops with SSE4.2 and a 256-entry look-up table, FRIES/math_utils.c:9-98).  Host code, header-only. */
#ifndef math_utils_h
#define math_utils_h
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define CEILING(x,y) ((x + y - 1) / y)
#define SIGN(x) ((x > 0) - (x < 0))
#define TRI_N(n)((n) * (n + 1) / 2)
#define I_J_TO_TRI_NODIAG(i, j)(TRI_N(j - 1) + i)
#define I_J_TO_TRI_WDIAG(i, j)(TRI_N(j) + i)

static inline uint64_t fries_word_of(const uint8_t *bit_str, uint8_t n_bytes) { uint64_t w = 0; memcpy(&w, bit_str, n_bytes > 8 ? 8 : n_bytes); return w; }

/* positions of the 1 bits, ascending; returns their number (math_utils.c:62-98) */
static inline uint8_t find_bits(const uint8_t *bit_str, uint8_t *bits, uint8_t n_bytes) {
    uint8_t n = 0;
    for (uint8_t b = 0; b < n_bytes; b++) for (uint8_t k = 0; k < 8; k++) if (bit_str[b] & (1u << k)) bits[n++] = (uint8_t)(8 * b + k);
    return n;
}
/* positions where the two strings differ, ascending; their number, or UINT8_MAX when they differ in more than four places
 * (math_utils.c:100-143) */
static inline uint8_t find_diff_bits(const uint8_t *str1, const uint8_t *str2, uint8_t *bits, uint8_t n_bytes) {
    uint8_t n = 0;
    for (uint8_t b = 0; b < n_bytes; b++) {
        const uint8_t x = (uint8_t)(str1[b] ^ str2[b]);
        for (uint8_t k = 0; k < 8; k++) if (x & (1u << k)) { if (n == 4) return UINT8_MAX; bits[n++] = (uint8_t)(8 * b + k); }
    }
    return n;
}
/* number of 1 bits strictly between positions a and b (math_utils.c:9-58) */
static inline unsigned int bits_between(uint8_t *bit_str, uint8_t a, uint8_t b) {
    uint8_t lo = a < b ? a : b, hi = a < b ? b : a;
    unsigned int n = 0;
    for (unsigned k = lo + 1u; k < hi; k++) n += (bit_str[k / 8] >> (k % 8)) & 1u;
    return n;
}
/* sorted-list edits used by the excitation bookkeeping (math_utils.c:147-196) */
static inline void new_sorted(uint8_t *orig_list, uint8_t *new_list, uint8_t length, uint8_t del_idx, uint8_t new_el) {
    uint8_t j = 0; int placed = 0;
    for (uint8_t i = 0; i < length; i++) {
        if (i == del_idx) continue;
        if (!placed && new_el < orig_list[i]) { new_list[j++] = new_el; placed = 1; }
        new_list[j++] = orig_list[i];
    }
    if (!placed) new_list[j++] = new_el;
}
static inline void repl_sorted(uint8_t *srt_list, uint8_t length, uint8_t del_idx, uint8_t new_el) {
    uint8_t tmp[256];
    new_sorted(srt_list, tmp, length, del_idx, new_el);
    memcpy(srt_list, tmp, length);
}
#endif
