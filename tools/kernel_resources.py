"""Static facts about the kernels of pt_engine.hip as hipcc builds them for gfx950: registers, spills, scratch, occupancy (the
compiler's -Rpass-analysis=kernel-resource-usage remarks) and the number of instructions by class from the disassembly.  CPU only.
usage: python tools/kernel_resources.py [EXTRA flags...]      e.g.  python tools/kernel_resources.py -DPT_ARITH_FMA=1 -ffp-contract=fast"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mygpuraytracer_amd", "csrc")
OPT = "-Os -fno-unroll-loops -fno-slp-vectorize -mllvm -disable-machine-licm".split()


def main(extra, src="pt_engine.hip"):
    base = ["hipcc", "--offload-arch=gfx950", "-std=c++17", *OPT, "-fPIC", "-ffp-contract=off", *extra, "-Wno-unused-function", "-Wno-unused-value"]
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(base + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", src, "-o", os.path.join(td, "k.s")],
                           cwd=CSRC, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            sys.exit(1)
        res, cur = collections.OrderedDict(), None
        for ln in r.stderr.splitlines():
            m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", ln)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = m.group(2); res[cur] = {}
            elif cur:
                res[cur][m.group(1)] = m.group(2)
        dis = open(os.path.join(td, "k.s")).read()
        counts, cur = {}, None
        for ln in dis.splitlines():
            m = re.match(r"^(_Z\S+|k_\w+):\s", ln)
            if m:
                cur = m.group(1); counts[cur] = collections.Counter(); continue
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
            m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
            if cur and m:
                op = m.group(1)
                k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
                counts[cur][k] += 1
                if op.startswith(("v_fma", "v_fmac", "v_pk_fma")): counts[cur]["fma"] += 1
                if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac")): counts[cur]["muladd"] += 1
        dm = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
        for k, v in res.items():
            c = counts.get(k, {})
            name = dm(k).replace("(anonymous namespace)::", "").replace("(BounceParams)", "")
            print("%-44s vgpr %3s occ %s sspill %3s vspill %s scratch %s | valu %5d (fma %4d, mul/add %4d) salu %5d lds %4d vmem %4d" % (
                name[:44], v.get("VGPRs"), v.get("Occupancy [waves/SIMD]"), v.get("SGPRs Spill"), v.get("VGPRs Spill", "?"), v.get("ScratchSize [bytes/lane]"),
                c.get("valu", 0), c.get("fma", 0), c.get("muladd", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0)))


if __name__ == "__main__":
    main(sys.argv[1:])
