/*
 * oracle/ref_shim/mfcc_f32_ref_shim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Calls the reference's OWN table builders of MFCC variant D (firmware/src/audio/mfcc.c, compiled from where it lies into
 * oracle/_ref/libmfcc_f32_ref.so by `make -C oracle f32ref`):
 *
 *   create_dct_matrix(input_length, coefficient_count)   mfcc.c:101-115
 *   create_mel_fbank(mfcc_t *)                            mfcc.c:117-171   on a hand-filled mfcc_t (mfcc.h:36-52)
 *
 * and flattens what they return into caller buffers. mfcc_create / mfcc_compute of the same object reference
 * arm_rfft_fast_init_f32 / arm_rfft_fast_f32, whose tables (arm_common_tables.c) are absent from the snapshot: those two
 * symbols stay unresolved and are never reached, so the library must be opened with lazy binding
 * (oracle_dl_open_lazy in oracle/ref_loader.c; Python's ctypes insists on RTLD_NOW).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mfcc.h"

void *oracle_host_roomy_malloc(size_t n) { return (malloc)(2 * n + 128); } /* see mfcc_f32_host_alloc.h */

/* out: [coefficient_count][input_length] float32, the reference's row order */
int f32ref_dct_matrix(int input_length, int coefficient_count, float *out)
{
	float *m = create_dct_matrix(input_length, coefficient_count);
	if (!m) return -1;
	memcpy(out, m, sizeof(float) * (size_t)input_length * (size_t)coefficient_count);
	free(m);
	return 0;
}

/* first / last: [26]; weights: the rows back to back (row b holds last[b] - first[b] + 1 values), at most cap floats;
 * returns the number of weights written, or -1 */
int f32ref_mel_fbank(int frame_len_padded, int32_t *first, int32_t *last, float *weights, int cap)
{
	mfcc_t m;
	memset(&m, 0, sizeof(m));
	m.frame_len_padded = frame_len_padded;
	m.fbank_filter_first = first;
	m.fbank_filter_last = last;
	float **rows = create_mel_fbank(&m);
	if (!rows) return -1;
	int pos = 0;
	for (int b = 0; b < NUM_FBANK_BINS; b++)
	{
		const int n = last[b] - first[b] + 1;
		if (first[b] < 0 || n < 0 || pos + n > cap) return -1;
		memcpy(weights + pos, rows[b], sizeof(float) * (size_t)n);
		pos += n;
	}
	return pos; /* the rows are left to the process: mfcc_delete would need a whole mfcc_t */
}
