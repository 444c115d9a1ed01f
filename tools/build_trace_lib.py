"""Build tools/libbosship_tr.so: a copy of the library whose chain kernels (diag / solve / column update / bulk
update) stamp s_memrealtime at entry and exit of workgroup 0 into a device array, read back by
tools/chain_trace.py.  The product sources are not modified; the instrumented copy lives in a temp dir."""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="bosship_tr_")
os.makedirs(os.path.join(tmp, "boss.jl_amd"))
shutil.copytree(os.path.join(ROOT, "boss.jl_amd", "csrc"), os.path.join(tmp, "boss.jl_amd", "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
p = os.path.join(tmp, "boss.jl_amd", "csrc", "potrf.hpp")
s = open(p).read()
s = s.replace("constexpr int DIAG_THREADS = 1024;",
              "__device__ unsigned long long g_tr[4096];\n__device__ unsigned int g_tri;\n"
              "#define TR_BEGIN(code) unsigned int tr_slot_ = 0; if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) { "
              "tr_slot_ = atomicAdd(&g_tri, 3u); if (tr_slot_ + 3 <= 4096) { g_tr[tr_slot_] = (code); "
              "g_tr[tr_slot_ + 1] = __builtin_amdgcn_s_memrealtime(); } }\n"
              "#define TR_END() if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0 && tr_slot_ + 3 <= 4096) "
              "g_tr[tr_slot_ + 2] = __builtin_amdgcn_s_memrealtime();\nconstexpr int DIAG_THREADS = 1024;", 1)
s = s.replace("    extern __shared__ double smem[];\n    double* D = smem;", "    TR_BEGIN(1);\n    extern __shared__ double smem[];\n    double* D = smem;", 1)
i = s.index("    // ---- write L: 16-byte stores")
j = s.index("\n}\n", i)
s = s[:j] + "\n    __syncthreads();\n    TR_END();" + s[j:]
for code, head in ((2, "__global__ __launch_bounds__(64) void potrf_trsm_kernel("),
                   (3, "__global__ __launch_bounds__(256, 2) void potrf_colupd_kernel("),
                   (4, "__global__ __launch_bounds__(256, 2) void potrf_syrk_kernel(")):
    i = s.index(head)
    k = s.index("{\n", i) + 2
    s = s[:k] + f"    TR_BEGIN({code});\n" + s[k:]
    j = s.index("\n}\n", k)
    s = s[:j] + "\n    TR_END();" + s[j:]
open(p, "w").write(s)
p = os.path.join(tmp, "boss.jl_amd", "csrc", "bosship.hip")
s = open(p).read() + '''
extern "C" int boss_debug_trace(unsigned long long* out, unsigned int* n, int reset) {
    unsigned int cnt = 0;
    if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(boss::g_tri), 4) != hipSuccess) return 1;
    if (cnt > 4096) cnt = 4096;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(boss::g_tr), 8 * cnt) != hipSuccess) return 1;
    *n = cnt;
    if (reset) { unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_tri), &z, 4); }
    return 0;
}
'''
open(p, "w").write(s)
out = os.path.join(ROOT, "tools", "libbosship_tr.so")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + entry.HIPCC_FLAGS + ["-o", out, p]
subprocess.check_call(cmd)
shutil.rmtree(tmp)
print("built", out)
