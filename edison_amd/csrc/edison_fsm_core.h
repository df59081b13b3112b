/*
 * edison_fsm_core.h -- one step of edisonFSM (firmware/src/app.c:727-928, without the LED strip), written once for the host
 * (legacy.c: edison_fsm_step) and for the device (edison_stream.hip: the stream's output stage walks the inferences of a push
 * through it; cnn_mfma_kernels.hip: the one-launch microphone push). The keyword roles -- EDI_WAKEWORD (app.c:50), ediLocations /
 * ediValues (app.c:135-147), resolved by name against the keyword list as the EDI_RESET state does (app.c:770-784) -- arrive as
 * an index and two bit masks over the ten classes (edison_fsm_roles, legacy.c).
 */
#ifndef EDISON_FSM_CORE_H
#define EDISON_FSM_CORE_H
#include "../../include/edison_hip.h"

#ifdef __HIPCC__
#define ED_FSM_FN __host__ __device__ static inline
#else
#define ED_FSM_FN static inline
#endif
#define ED_FSM_LOC_TIMEOUT_MS 5000u /* EDI_LOC_TIMEOUT, app.c:48 */

typedef struct
{
	int32_t wake_idx;               /* class index of the wake word, -1 if the keyword list has none */
	uint32_t loc_mask, val_mask;    /* bit c: class c is a location / a value */
} ed_fsm_roles_t;

/* hit: the filtered maximum exceeded TRUE_THRESHOLD (app.c:346); pred_idx: its class (arm_max_f32). Returns the new state,
 * -1 for a state that does not exist. */
ED_FSM_FN int ed_fsm_step_core(edison_fsm *f, int hit, uint32_t pred_idx, uint32_t dt_us, const ed_fsm_roles_t *roles)
{
	int next = f->state;
	const int is_loc = pred_idx < 32u && ((roles->loc_mask >> pred_idx) & 1u), is_val = pred_idx < 32u && ((roles->val_mask >> pred_idx) & 1u);
	switch (f->state)
	{
	case EDISON_FSM_RESET: /* app.c:766-791 */
		f->wake_idx = roles->wake_idx;
		next = EDISON_FSM_IDLE;
		break;
	case EDISON_FSM_IDLE: /* app.c:793-800 */
		if (hit && (int)pred_idx == f->wake_idx) { f->hot_timeout_ms = 0; next = EDISON_FSM_HOT; }
		break;
	case EDISON_FSM_HOT: /* app.c:801-824; the firmware's truncating `hotTimeout += dt/1000` */
		f->hot_timeout_ms += dt_us / 1000u;
		if (hit && is_loc) { f->loc_idx = (int)pred_idx; f->hot_timeout_ms = 0; next = EDISON_FSM_LOC; }
		if (f->hot_timeout_ms > ED_FSM_LOC_TIMEOUT_MS) next = EDISON_FSM_IDLE;
		break;
	case EDISON_FSM_LOC: /* app.c:826-848: a value found at the very call that times out is dropped */
		f->hot_timeout_ms += dt_us / 1000u;
		if (hit && is_val) { f->val_idx = (int)pred_idx; next = EDISON_FSM_SET; }
		if (f->hot_timeout_ms > ED_FSM_LOC_TIMEOUT_MS) next = EDISON_FSM_IDLE;
		break;
	case EDISON_FSM_SET: /* app.c:850-872: "set location to required value", then idle */
		f->last_loc = f->loc_idx;
		f->last_val = f->val_idx;
		f->commands++;
		next = EDISON_FSM_IDLE;
		break;
	default:
		return -1;
	}
	f->state = next;
	return next;
}
#endif
