/*
 * oracle/ref_shim/mfcc_f32_host_alloc.h -- TEST INFRASTRUCTURE. Force-included (gcc -include) in front of the reference's
 * firmware/src/audio/mfcc.c when oracle/Makefile builds oracle/_ref/libmfcc_f32_ref.so for a 64-bit host.
 *
 * Why: that file is written for the 32-bit MCU. create_mel_fbank sizes its array of 26 row POINTERS as
 * sizeof(float) * 26 (mfcc.c:129: 104 bytes, 208 needed where a pointer is 8 bytes), and mfcc_malloc clears
 * sizeof(mfcc_t) bytes of EVERY block whatever its size (mfcc.c:34-39). On x86-64 both run past the block. Every request
 * of that translation unit is therefore served with room to spare (2 n + 128 bytes); no value the reference computes
 * changes, only where its overruns land. Nothing else of the reference is touched.
 */
#ifndef ORACLE_MFCC_F32_HOST_ALLOC_H
#define ORACLE_MFCC_F32_HOST_ALLOC_H
#include <stdlib.h>
void *oracle_host_roomy_malloc(size_t n);
#define malloc(n) oracle_host_roomy_malloc(n)
#endif
