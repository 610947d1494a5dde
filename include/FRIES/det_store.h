/* FRIES/det_store.h:23-55 for the MI355X build (host, header-only): bit i of a determinant lives in byte i / 8, bit i % 8. */
#ifndef det_store_h
#define det_store_h
#include <stdint.h>
#include <stdio.h>
#include "math_utils.h"
static inline int read_bit(const uint8_t *bit_str, uint8_t bit_idx) { return !(!(bit_str[bit_idx / 8] & (1 << (bit_idx % 8)))); }
static inline void zero_bit(uint8_t *bit_str, uint8_t bit_idx) { bit_str[bit_idx / 8] &= (uint8_t)~(1u << (bit_idx % 8)); }
static inline void set_bit(uint8_t *bit_str, uint8_t bit_idx) { bit_str[bit_idx / 8] |= (uint8_t)(1u << (bit_idx % 8)); }
static inline void print_str(uint8_t *bit_str, uint8_t n_bytes, char *out_str) { for (uint8_t b = 0; b < n_bytes; b++) sprintf(&out_str[2 * b], "%02x", bit_str[b]); }
#endif
