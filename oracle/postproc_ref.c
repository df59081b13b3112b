/*
 * oracle/postproc_ref.c -- TEST INFRASTRUCTURE. The firmware's post-processing of the network output in
 * continuous mode (firmware/src/app.c:332-356): cast to float, moving average in DOUBLE arithmetic (the constants
 * NET_OUT_MOVING_AVG_ALPHA and 1.0 are doubles, app.c:38,342) rounded to float on the store, arm_max_f32 (first
 * maximum: CMSIS StatisticsFunctions/arm_max_f32.c scalar branch, strict `out < maxVal`), TRUE_THRESHOLD compare.
 * Parity: restated from the source; the reference holds no vectors for it (parity unpinned) -- except the class choice, which is
 * checked against CMSIS-DSP's arm_max_f32 compiled from the reference (tests/test_oracle_refpins.py).
 */
#include <stdint.h>

#include "oracle.h"

/* no fused multiply-add: the Cortex-M4 has no double FPU, every double operation is a separately rounded call */
__attribute__((optimize("fp-contract=off")))
void oracle_output_filter(const int8_t *soft, int64_t n, double alpha, double threshold, float *state, float *filt,
                          int32_t *likely, int32_t *spotted)
{
	const double one_minus_alpha = 1.0 - alpha;
	for (int64_t i = 0; i < n; i++)
	{
		for (int c = 0; c < 10; c++)
		{
			const float x = (float)soft[i * 10 + c];
			volatile double a = alpha * (double)state[c];
			volatile double b = one_minus_alpha * (double)x;
			state[c] = (float)(a + b);
			if (filt) filt[i * 10 + c] = state[c];
		}
		float best = state[0];
		int idx = 0;
		for (int c = 1; c < 10; c++)
			if (best < state[c]) { best = state[c]; idx = c; }
		if (likely) likely[i] = idx;
		if (spotted) spotted[i] = ((double)best > threshold) ? idx : -1;
	}
}
