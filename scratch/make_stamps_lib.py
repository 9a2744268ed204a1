"""build scratch/_lib_stamps.so: the product library with wall_clock64 phase stamps inside bottleneck_fwd_kernel (workgroup 7, thread 0)"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "hid-vae_amd/csrc/rq.hip")).read()
a = src.index("template <int MODE>\n__global__ __launch_bounds__(64 * BN_WAVES) void bottleneck_fwd_kernel")
b = src.index("// ------------------------------------------------------------------------------------------------\n// backward: recompute")
body = src[a:b]
anchors = [
    ("    const FwdArgs &a = b.rq;\n", 0, "after"),
    ("    for (int i = 0; i < a.L; i++) stage_codes(lds + i * lvl_floats, a, i, 0);\n    __syncthreads();\n", 1, "after"),
    ("    bneck_layer<BN_WAVES>(b.W2, b.N2, b.K2, HA, true, b.pre2, b.h2, HB, item, in_range, wave, lane);\n    __syncthreads();\n", 2, "after"),
    ("    bneck_layer<BN_WAVES>(b.W3, D, b.N2, HB, false, nullptr, b.y_out, HY, item, in_range, wave, lane);\n    __syncthreads();\n", 3, "after"),
    ("    rq_level_loop<MODE, true, true, true, BN_WAVES>(a, lds, cand_d, cand_i, phase, wave, it, q, item, valid, r, esum, loss, tuple);\n", 4, "after"),
    ("    bneck_layer<BN_WAVES>(b.Wd0, b.Nd0, D, HY, true, b.pre_d0, b.d0, HA, item, in_range, wave, lane);\n    __syncthreads();\n", 5, "after"),
    ("    bneck_layer<BN_WAVES>(b.Wd1, b.Nd1, b.Nd0, HA, true, b.pre_d1, b.d1, nullptr, item, in_range, wave, lane);\n", 6, "after"),
]
for text, idx, _ in anchors:
    assert body.count(text) == 1, text
    body = body.replace(text, text + f"    STAMP({idx});\n", 1)
# end stamp: before the final closing brace of the kernel
k = body.rindex("}\n")
body = body[:k] + "    STAMP(7);\n" + body[k:]
pre = ("__device__ unsigned long long g_stamps[16];\n#define STAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)\n")
out = src[:a] + pre + body + src[b:]
out += ('\nextern "C" int hidvae_debug_stamps(unsigned long long *out) {\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long));\n}\n')
os.makedirs(os.path.join(R, "scratch/_stamps"), exist_ok=True)
p = os.path.join(R, "hid-vae_amd/csrc/_rq_stamps_tmp.hip")
open(p, "w").write(out)
try:
    obj = os.path.join(R, "scratch/_stamps/rq_stamps.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value",
                           "-Wno-pass-failed", "-x", "hip", "-c", p, "-o", obj])
finally:
    os.remove(p)
objs = [os.path.join(R, "hid-vae_amd/build", f) for f in os.listdir(os.path.join(R, "hid-vae_amd/build")) if f.endswith(".o") and not f.startswith("rq.hip")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "scratch/_lib_stamps.so"), obj] + objs)
print("ok")
