#!/usr/bin/env python3
"""Generate tools/ablate/mfcc_kernels_stamp.hip: the MFCC kernel with s_memtime stamps at phase boundaries
(diagnostic build only; per-wave cycle sums per phase go to a debug buffer set with ed_set_debug_buffer)."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, 'edison_amd/csrc/mfcc_kernels.hip')).read()
stamp = '''
#define ED_NPH 12
__device__ unsigned long long *g_ed_dbg = nullptr;
extern "C" void ed_set_debug_buffer(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ed_dbg), &p, sizeof(p)); }
__device__ __forceinline__ unsigned long long ed_now()
{
	unsigned long long t;
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
	__builtin_amdgcn_sched_barrier(0);
	return t;
}
#define ED_STAMP(i) { unsigned long long n_ = ed_now(); ph[i] += n_ - tlast; tlast = n_; }
'''
s = s.replace('__device__ __forceinline__ void ed_wave_sync()', stamp + '__device__ __forceinline__ void ed_wave_sync()')
s = s.replace('	for (; f < n_frames; f += stride)\n	{', '	unsigned long long ph[ED_NPH]; for (int i_ = 0; i_ < ED_NPH; i_++) ph[i_] = 0;\n	unsigned long long tlast = ed_now();\n	for (; f < n_frames; f += stride)\n	{\n		ED_STAMP(11)')
marks = ['		/* ---- 2a. pass 1', '		/* transpose 1:', '		/* ---- 2b. pass 2', '#if ED_T2_LDS\n		/* transpose 2 through', '		/* ---- 2c. pass 3',
         '		/* ---- 3. real-FFT split.', '		/* ---- 4. spectrum to LDS', '		/* ---- 5. mel filterbank', '		/* ---- 6. DCT-II', '		/* ---- 7. store */']
for i, m in enumerate(marks):
    assert s.count(m) == 1, m
    s = s.replace(m, '		ED_STAMP(%d)\n%s' % (i, m)) if not m.startswith('#if') else s.replace(m, '		ED_STAMP(%d)\n%s' % (i, m))
tail = '''				args.feat[(int64_t)f * args.n_coef + lane] = (int8_t)__float2int_rn(q);
			}
		}
	}
}'''
assert s.count(tail) == 1
s = s.replace(tail, '''				args.feat[(int64_t)f * args.n_coef + lane] = (int8_t)__float2int_rn(q);
			}
		}
		ED_STAMP(10)
	}
	if (!STAGES && g_ed_dbg && lane == 0)
	{
		unsigned long long *dbg = g_ed_dbg + (size_t)(blockIdx.x * ED_WPB + wave) * ED_NPH;
		for (int i_ = 0; i_ < ED_NPH; i_++) dbg[i_] = ph[i_];
	}
}''')
open(os.path.join(ROOT, 'tools/ablate/mfcc_kernels_stamp.hip'), 'w').write(s)
print("ok")
