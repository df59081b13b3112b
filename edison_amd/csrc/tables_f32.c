/*
 * tables_f32.c -- host-side construction of the constant tables of MFCC variant D, the firmware's float32
 * ML-KWS feature extractor (firmware/src/audio/mfcc.c). Float arithmetic where the firmware uses float:
 *   window        mfcc.c:66-68    0.5f - 0.5f cosf(2 pi i / frame_len)
 *   mel filters   mfcc.c:119-172  26 triangles between 20 and 4000 Hz, weights linear in MelScale (mfcc.h:53-55)
 *   DCT matrix    mfcc.c:102-117  sqrt(2/26) cosf(pi/26 (n + 0.5) k)
 * plus the twiddles of the device FFT (the firmware calls arm_rfft_fast_f32; its tables are not in the snapshot).
 */
/* Every float operation below is rounded on its own, as the C source of mfcc.c says (the firmware compiles it at -O0,
 * firmware/Makefile:42; the reference's file compiled on this host agrees: tests/test_oracle_refpins.py compares the tables
 * with its create_mel_fbank / create_dct_matrix bit for bit). clang would otherwise fuse a * b + c into one fma. */
#pragma STDC FP_CONTRACT OFF

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

static float mel_scale_f(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

int ed_build_f32_tables(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits, float preempha,
                        ed_f32_tables_t *out, char *err, size_t err_cap)
{
	memset(out, 0, sizeof(*out));
	if (num_mfcc_features < 1 || num_mfcc_features > ED_F32_NUM_FBANK || feature_offset < 0 ||
	    feature_offset >= num_mfcc_features || frame_len < 2 || mfcc_dec_bits < 0 || mfcc_dec_bits > 30)
	{
		if (err) snprintf(err, err_cap, "mfcc_create: bad arguments");
		return EDISON_E_ARGUMENT;
	}
	int padded = 1, log2p = 0;
	while (padded < frame_len) { padded <<= 1; log2p++; } /* powf(2, ceilf(logf(n)/logf(2))), mfcc.c:58 */
	if (padded > ED_F32_MAX_FRAME || padded < 128)
	{
		if (err) snprintf(err, err_cap, "mfcc_create: frame_len %d pads to %d, this path handles 128..%d", frame_len, padded,
		                  ED_F32_MAX_FRAME);
		return EDISON_E_NO_IMPL;
	}
	out->n_features = num_mfcc_features; out->offset = feature_offset; out->frame_len = frame_len;
	out->padded = padded; out->log2p = log2p; out->dec_bits = mfcc_dec_bits;
	out->preempha = preempha;
	out->scale = (float)(1u << mfcc_dec_bits);
	for (int i = 0; i < frame_len; i++)
		out->window[i] = 0.5f - 0.5f * cosf((float)6.283185307179586476925286766559005 * ((float)i) / (frame_len));
	for (int k = 0; k < padded / 2; k++)
	{
		const double a = -2.0 * M_PI * (double)k / (double)padded;
		out->tw[k][0] = (float)cos(a); out->tw[k][1] = (float)sin(a);
	}
	const int nbins = padded / 2;
	const float bin_width = 16000.0f / padded;
	const float lo = mel_scale_f(20), hi = mel_scale_f(4000);
	const float delta = (hi - lo) / (ED_F32_NUM_FBANK + 1);
	int pos = 0;
	for (int b = 0; b < ED_F32_NUM_FBANK; b++)
	{
		const float left = lo + b * delta, center = lo + (b + 1) * delta, right = lo + (b + 2) * delta;
		int first = -1, last = -1;
		out->mel_off[b] = pos;
		for (int i = 0; i < nbins; i++)
		{
			const float mel = mel_scale_f(bin_width * i);
			if (mel > left && mel < right)
			{
				const float w = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
				if (first == -1) first = i;
				last = i;
				if (pos >= ED_F32_MAX_W)
				{
					if (err) snprintf(err, err_cap, "mfcc_create: mel filterbank too wide for this path");
					return EDISON_E_NO_IMPL;
				}
				out->mel_w[pos++] = w; /* inside (left, right) every bin has a weight, so the run is contiguous */
			}
		}
		out->mel_first[b] = first; out->mel_last[b] = last;
	}
	float *dct = create_dct_matrix(ED_F32_NUM_FBANK, num_mfcc_features);
	if (!dct) return EDISON_E_NO_MEMORY;
	memcpy(out->dct, dct, sizeof(float) * ED_F32_NUM_FBANK * (size_t)num_mfcc_features);
	free(dct);
	return EDISON_OK;
}

/* mfcc.c:101-115 under its own name: M[k][n] = sqrt(2/N) cos(pi/N (n + 1/2) k), all in float32 */
float *create_dct_matrix(int32_t input_length, int32_t coefficient_count)
{
	if (input_length < 1 || coefficient_count < 1) return NULL;
	float *m = (float *)malloc(sizeof(float) * (size_t)input_length * (size_t)coefficient_count);
	if (!m) return NULL;
	const float normalizer = sqrtf(2.0f / (float)input_length);
	for (int32_t k = 0; k < coefficient_count; k++)
		for (int32_t n = 0; n < input_length; n++)
			m[k * input_length + n] = normalizer * cosf(((float)3.14159265358979323846264338327950288) / input_length * (n + 0.5f) * k);
	return m;
}
