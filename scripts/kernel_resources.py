"""Per-kernel resource record of the product library's device code: registers, scratch (spill) bytes, LDS and the
occupancy the compiler derives from them, one line per kernel, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (no GPU needed).  Usage: python scripts/kernel_resources.py > profiles/r3/kernel_resources.txt"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "opencl-ray-tracer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", "rt_kernels.hip"]
err = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        name = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("(RenderParams)", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
print("%-58s %5s %5s %5s %9s %7s %6s %6s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch B", "LDS B", "occ", "vspill", "sspill"))
for r in rows:
    print("%-58s %5s %5s %5s %9s %7s %6s %6s %6s" % (r["name"][:58], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]"),
          r.get("VGPRs Spill"), r.get("SGPRs Spill")))
print("\nocc = waves per SIMD the register/LDS budget allows; scratch B = private memory per lane (register spills and the\n"
      "counting build's counter arrays); k_render<STATS, MESH, BOXTREE, MULTI>; strict_impl / tolerant_impl = the two arithmetic builds.")
