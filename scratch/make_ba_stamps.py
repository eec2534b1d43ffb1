"""ba.hip -> scratch/abvar/ba_stamps.hip: s_memtime stamps at the phase boundaries of ba_solve_dense_kernel (read back by
scratch/dense_stamps.py through the variant library's vipe_dbg_dn_stamps)"""
s = open("vipe_amd/csrc/ba.hip").read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)
rep("constexpr int DN_T = 512;\n", "constexpr int DN_T = 512;\n__device__ unsigned long long g_dn_stamps[1024];\n#define STAMP(i) do { if (t == 0 && (i) < 1024) g_dn_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)\n")
rep("  if (t == 0) *failp = 0;\n  // ---- the matrix (rows 0..n", "  STAMP(0);\n  if (t == 0) *failp = 0;\n  // ---- the matrix (rows 0..n")
import re, sys
# generic markers placed in the source as comments: // @stamp N  or  // @stampk N  (8 + 8 * kb + N)
s = re.sub(r"// @stampk (\d+)", lambda m: f"STAMP(8 + 8 * kb + {m.group(1)});", s)
s = re.sub(r"// @stamp (\d+)", lambda m: f"STAMP({m.group(1)});", s)
s += """
extern "C" VIPE_EXPORT int vipe_dbg_dn_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dn_stamps), sizeof(unsigned long long) * 1024);
}
"""
open("scratch/abvar/ba_stamps.hip", "w").write(s)
print("stamps:", s.count("STAMP("))
