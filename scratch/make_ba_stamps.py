"""ba.hip -> scratch/abvar/ba_stamps.hip: s_memtime stamps at the phase boundaries of ba_solve_dense_kernel (read back by
scratch/dense_stamps.py through the variant library's vipe_dbg_dn_stamps)"""
s = open("vipe_amd/csrc/ba.hip").read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)
rep("constexpr int AM_ROWS = 48;", "__device__ unsigned long long g_dn_stamps[4096];\nconstexpr int AM_ROWS = 48;")
rep("constexpr int DN_T = 512;\n", "constexpr int DN_T = 512;\n#define STAMP(i) do { if (t == 0 && (i) < 1024) g_dn_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)\n")
import re, sys
# generic markers placed in the source as comments: // @stamp N  or  // @stampk N  (8 + 8 * kb + N)
s = re.sub(r"// @wstampk (\d+)", lambda m: f"if (twave >= 0 && lane == 0 && 8 * kb + {m.group(1)} < 256) g_dn_stamps[512 + 256 * twave + 8 * kb + {m.group(1)}] = __builtin_amdgcn_s_memtime();", s)
s = re.sub(r"// @kstampc (\d+)", lambda m: f"if (threadIdx.x == 0 && blockIdx.x == 3) {{ g_dn_stamps[3400 + 32 * (blockIdx.y & 15) + 6 * (cb / WK_CH1) + {m.group(1)}] = __builtin_amdgcn_s_memtime(); g_dn_stamps[3400 + 32 * (blockIdx.y & 15) + 31] = deg_all; }}", s)
s = re.sub(r"// @kstamp (\d+)", lambda m: f"if (threadIdx.x == 0 && blockIdx.x == 3) g_dn_stamps[3400 + 32 * (blockIdx.y & 15) + {m.group(1)}] = __builtin_amdgcn_s_memtime();", s)
s = re.sub(r"// @astamp (\d+)", lambda m: f"if (threadIdx.x == 0 && blockIdx.x == 3 && (blockIdx.y == 24 || blockIdx.y == 2)) g_dn_stamps[3300 + 16 * (blockIdx.y == 24) + {m.group(1)}] = __builtin_amdgcn_s_memtime();", s)
s = re.sub(r"// @bwave (\d+)", lambda m: f"if ((threadIdx.x & 63) == 0) g_dn_stamps[3000 + {m.group(1)} + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();", s)
s = re.sub(r"// @bstamp (\d+)", lambda m: f"if (threadIdx.x == 0) g_dn_stamps[3000 + {m.group(1)}] = __builtin_amdgcn_s_memtime();", s)
s = re.sub(r"// @stampk (\d+)", lambda m: f"STAMP(8 + 8 * kb + {m.group(1)});", s)
s = re.sub(r"// @stamp (\d+)", lambda m: f"STAMP({m.group(1)});", s)
s += """
extern "C" VIPE_EXPORT int vipe_dbg_dn_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dn_stamps), sizeof(unsigned long long) * 4096);
}
"""
open("scratch/abvar/ba_stamps.hip", "w").write(s)
print("stamps:", s.count("STAMP("))
