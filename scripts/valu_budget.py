"""Instruction budget of the headline render kernel (strict k_render<false,false>): where its VALU wave-instructions go, by
phase of the render loop, with the active lanes of each phase.

  python scripts/valu_budget.py run [spp]        on the GPU box, with the instrumented variant
                                                 (make -C opencl-ray-tracer_amd/csrc ../variants/phase_stats.so):
                                                 wave-level executions and active lanes of every phase -> gpurun_out/phase_counts.json
  python scripts/valu_budget.py table [json]     anywhere: static instruction counts per phase from the device assembly with
                                                 line tables (make -C opencl-ray-tracer_amd/csrc rt_kernels_lines.s; same code as
                                                 the shipped build), times those executions

Static attribution: every instruction of the kernel carries the chain of inlined source lines it comes from; the outermost
frame below the kernel's own body names the phase.  Executions are counted per WAVE (an instruction issues once per wave
whatever the number of active lanes); lanes = mean share of the 64 lanes active when the phase runs.
"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opencl-ray-tracer_amd", "csrc")
PHASES = ["iterations", "sphere tests past the early-out", "exact sphere loops", "surface + material", "rejection-loop rounds",
          "sky texels", "exact sky texels", "sample ends", "camera rays", "div3 fallbacks", "div_by_const fallbacks",
          "winner roots", "disk-loop rounds", "Lambertian scatters", "metal scatters", "dielectric scatters"]


def run(spp):
    os.environ["WBRT_LIB"] = os.path.join(ROOT, "opencl-ray-tracer_amd", "variants", "phase_stats.so")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    from util import pkg
    scene, cam = pkg.scene_readme(), pkg.camera_readme(1280, spp, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0)
    ctx.upload_scene(scene)
    ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    ctx.tune("lpt", 0)  # one launch, no pilot: the counts do not depend on the launch order
    out = (C.c_ulonglong * 32)()
    pkg.lib().wbrt_dbg_phases_(out, 1)
    ctx.render_frame(cam)
    pkg.lib().wbrt_dbg_phases_(out, 1)
    ctx.close()
    res = {"spp": spp, "width": cam.width, "height": cam.height, "spheres": 5,
           "phases": {PHASES[k]: {"wave_executions": int(out[2 * k]), "lanes": int(out[2 * k + 1])} for k in range(len(PHASES))}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "phase_counts.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def source_ranges():
    """name -> (first line, last line) of the parts of rt_render.inc, from anchors in its text."""
    src = open(os.path.join(CSRC, "rt_render.inc")).read().split("\n")

    def find(text, after=0):
        for i in range(after, len(src)):
            if text in src[i]:
                return i + 1
        raise KeyError(text)
    a = {
        "vec": (1, find("template <bool ON> struct Counters;") - 1),
        "rng": (find("struct Rng"), find("// ---------------------------------------------- portable transcendental math") - 1),
        "sky_exact": (find("// ---------------------------------------------- portable transcendental math"), find("// IEEE-correct num/den for three numerators") - 1),
        "div": (find("// IEEE-correct num/den for three numerators"), find("// atan(a) for a in [0,1]") - 1),
        "sky_fast": (find("// atan(a) for a in [0,1]"), find("// Closest sphere without per-sphere exact square roots") - 1),
        "scan": (find("// Closest sphere without per-sphere exact square roots"), find("// Tolerant build: the reference's loop over all spheres") - 2),
        "exact": (find("__device__ __forceinline__ void spheres_exact"), find("// Wave-private LDS staging area") - 1),
        "dispatch_sphere": (find("__device__ __forceinline__ void closest_sphere"), find("// gpu_kernel.cl:182-196") - 1),
        "rej": (find("// gpu_kernel.cl:182-196"), find("// gpu_kernel.cl:209-212") - 1),
        "reflect": (find("// gpu_kernel.cl:209-212"), find("// Pixel-stream state of one lane.") - 1),
        "cam": (find("// gpu_kernel.cl:626-629 + :559-578"), find("// Maps a launch-wide work index") - 1),
        "locate": (find("// Maps a launch-wide work index"), find("// gpu_kernel.cl:342-347: the sky texel a ray") - 1),
        "dispatch_sky": (find("// gpu_kernel.cl:342-347: the sky texel a ray"), find("// Surface and material of the nearest hit") - 1),
        "shade": (find("// Surface and material of the nearest hit"), find("// MESH selects the loop shape") - 1),
        "kernel": (find("// MESH selects the loop shape"), find("// Render kernel for scenes without meshes, workgroup form") - 1),
    }
    marks = {
        "scan_ballot": find("if (!__ballot(!(disc < 0.0f))) return;"),
        "scan_test": find("__device__ __forceinline__ void test(F3 o, F3 d"),
        "scan_finish": find("__device__ __forceinline__ bool finish("),
        "scan_loop": find("__device__ __forceinline__ bool spheres_fast"),
        "loop_top": find("int iterations = 0, walk_cost = 0;"),
        "sample_end": find("// gpu_kernel.cl:632-644: clamp, NaN -> running sum, accumulate"),
        "loop_end": find("if (STATS)\n".strip(), find("else alive = false;")),
        "div3_fallback": find("q = num / den; } // operands outside the range"),
        "divc_fallback": find("q = x / den; } // (rare)"),
        "finish_root": (find("// the reference's own choice of root (gpu_kernel.cl:468-478)"), find("// the winner's numerator lies in [min1, min1 + 2 margins]") - 1),
        "disk": (find("for (;;) // gpu_kernel.cl:243-257"), find("org = org + (ld3(p.ddu) * dx + ld3(p.ddv) * dy);")),
        "lambert": (find("if (type == 0) // gpu_kernel.cl:398-413"), find("else if (type == 1) // gpu_kernel.cl:415-423") - 1),
        "metal": (find("else if (type == 1) // gpu_kernel.cl:415-423"), find("else if (type == 2) // gpu_kernel.cl:425-451") - 1),
        "dielectric": (find("else if (type == 2) // gpu_kernel.cl:425-451"), find("else if (type == 3) // gpu_kernel.cl:326-329") - 1),
    }
    return a, marks


def classify(chain, ranges, marks):
    """chain: [(file, line)] innermost first.  Returns the budget row of the instruction."""
    frames = []
    for f, line in chain:
        if not f.endswith("rt_render.inc"):
            frames.append("other")
            continue
        name = "other"
        for k, (lo, hi) in ranges.items():
            if lo <= line <= hi:
                name = k
        frames.append(name)
    outer_in = list(reversed(list(zip(frames, [l for _, l in chain]))))  # outermost first
    names = [n for n, _ in outer_in]
    lines = [l for _, l in outer_in]
    if marks["div3_fallback"] in lines:
        return "full IEEE divisions where div3's short sequence does not apply (rare)"
    if marks["divc_fallback"] in lines:
        return "full IEEE division where div_by_const's short sequence does not apply (rare)"
    # code of the kernel's prologue / epilogue (the first camera ray of a stream is a copy of the in-loop one)
    kernel_lines = [l for n, l in outer_in if n == "kernel"]
    if kernel_lines and (kernel_lines[0] < marks["loop_top"] or kernel_lines[0] > marks["loop_end"]):
        return "prologue / epilogue"
    # the phase: first frame below the kernel's own body that is not a dispatcher
    for idx, (n, line) in enumerate(outer_in):
        if n in ("kernel", "dispatch_sphere", "dispatch_sky", "other"):
            continue
        inner = names[idx:]
        if n == "scan":
            line = [l for nn, l in outer_in if nn == "scan"][-1]  # the innermost frame inside the scan code
            if marks["scan_test"] <= line < marks["scan_finish"]:
                return "sphere scan: to the discriminant + wave-level early-out (per sphere)" if line <= marks["scan_ballot"] else \
                       "sphere scan: roots, margins, ordered bounds (per sphere past the early-out)"
            if line >= marks["scan_loop"]:
                return "sphere scan: loop control, scalar loads (per sphere)"
            if marks["finish_root"][0] <= line <= marks["finish_root"][1]:
                return "sphere scan: the winner's root, IEEE sqrt and division (where a lane has a candidate)"
            return "sphere scan: begin / finish (range guard, final margin check)"
        if n == "exact":
            return "exact sphere loop (fallback)"
        if n == "shade":
            if "rej" in inner:
                return "rejection sampling: LCG draws" if "rng" in inner else "rejection sampling: point + test"
            if "div" in inner:
                return "surface + material: IEEE normalisations (div3 / unit)"
            sline = [l for nn, l in outer_in if nn == "shade"][-1]
            for key, label in (("lambert", "scatter: Lambertian (where a lane has one)"), ("metal", "scatter: metal (where a lane has one)"),
                               ("dielectric", "scatter: dielectric (where a lane has one)")):
                if marks[key][0] <= sline <= marks[key][1]:
                    return label
            return "surface + material: hit point, normal, material fetch, ray update"
        if n == "rej":
            return "rejection sampling: LCG draws" if "rng" in inner else "rejection sampling: point + test"
        if n == "sky_fast":
            return "sky texel (fast path)"
        if n == "sky_exact":
            return "sky texel (exact fallback)"
        if n == "cam":
            cline = [l for nn, l in outer_in if nn == "cam"][-1]
            if marks["disk"][0] <= cline <= marks["disk"][1]:
                return "camera ray: defocus disk (rounds of its rejection loop)"
            if "rng" in inner:
                return "camera ray: LCG draws"
            if "div" in inner:
                return "camera ray: IEEE divisions by width / height"
            return "camera ray: direction"
        if n == "locate":
            return "prologue / epilogue"
        return "k_render glue"
    # the kernel's own lines
    line = outer_in[0][1] if outer_in else 0
    for n, l in outer_in:
        if n == "kernel":
            line = l
    if line < marks["loop_top"] or line > marks["loop_end"]:
        return "prologue / epilogue"
    if line >= marks["sample_end"]:
        return "sample end: clamp, NaN quirk, accumulate"
    return "loop glue: depth check, outcome selects, sky modulate"


# measured issue cost of a VALU instruction with 6 waves per SIMD, cycles at 2.4 GHz (scripts/ubench/valu_rate.hip)
FAST_OPS = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32",
            "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32")


def op_cycles(op):
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    base = base.replace("_dpp", "").replace("_sdwa", "")
    if base.startswith(("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
        return 8.06, "transcendental"
    if base.startswith("v_pk_"):
        return 4.45, "packed"
    if base in FAST_OPS:
        return 2.6, "fma / mul / add / move / simple integer"
    return 4.2, "other (min / max / compare / select / convert / shift / 3-operand integer / multiply)"


def table(path, by_op=False):
    counts = json.load(open(path))
    ph = counts["phases"]
    S = counts.get("spheres", 5)
    ranges, marks = source_ranges()
    asm = open(os.path.join(CSRC, "rt_kernels_lines.s")).read().split("\n")
    start = next(i for i, l in enumerate(asm) if l.startswith("_ZN12_GLOBAL__N_111strict_impl8k_renderILb0ELb0ELb0ELb0EEEv12RenderParams:"))
    rows = {}
    chain = []
    loc = re.compile(r"([^\s;@\[\]]+):(\d+):\d+")
    for l in asm[start + 1:]:
        if "s_endpgm" in l:
            break
        if l.lstrip().startswith(".loc"):
            c = l.split(";", 1)[1] if ";" in l else ""
            chain = [(m.group(1), int(m.group(2))) for m in loc.finditer(c)]
            continue
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m or l.startswith(".") or l.startswith(";"):
            continue
        op = m.group(1)
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_nop", "s_buffer_load")) else \
               "smem" if op.startswith(("s_load", "s_buffer_load")) else "lds" if op.startswith("ds_") else \
               "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else None
        if kind is None:
            continue
        row = classify(chain, ranges, marks)
        rows.setdefault(row, {"valu": 0, "salu": 0, "smem": 0, "lds": 0, "vmem": 0, "trans": 0, "ops": {}})
        rows[row][kind] += 1
        if kind == "valu":
            rows[row]["ops"][op] = rows[row]["ops"].get(op, 0) + 1
        if op.startswith(("v_rcp_", "v_sqrt_", "v_rsq_")):
            rows[row]["trans"] += 1
    it = ph["iterations"]["wave_executions"]
    waves = counts["width"] * counts["height"] // 64

    def execs(row):
        if row.startswith("sphere scan: to the discriminant") or row.startswith("sphere scan: loop control"):
            return S * it, ph["iterations"]["lanes"] * S
        if row.startswith("sphere scan: roots"):
            p = ph["sphere tests past the early-out"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("sphere scan: begin") or row.startswith("loop glue") or row == "k_render glue":
            return it, ph["iterations"]["lanes"]
        for prefix, key in (("sphere scan: the winner's root", "winner roots"), ("camera ray: defocus disk", "disk-loop rounds"),
                            ("scatter: Lambertian", "Lambertian scatters"), ("scatter: metal", "metal scatters"), ("scatter: dielectric", "dielectric scatters")):
            if row.startswith(prefix):
                return ph[key]["wave_executions"], ph[key]["lanes"]
        if row.startswith("exact sphere"):
            p = ph["exact sphere loops"]
            return S * p["wave_executions"], S * p["lanes"]
        if row.startswith("rejection"):
            p = ph["rejection-loop rounds"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("surface"):
            p = ph["surface + material"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("sky texel (fast"):
            p = ph["sky texels"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("sky texel (exact"):
            p = ph["exact sky texels"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("sample end"):
            p = ph["sample ends"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("camera"):
            p = ph["camera rays"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("full IEEE divisions where div3"):
            p = ph["div3 fallbacks"]
            return p["wave_executions"], p["lanes"]
        if row.startswith("full IEEE division where div_by_const"):
            p = ph["div_by_const fallbacks"]
            return p["wave_executions"], p["lanes"]
        return waves, waves * 64
    if by_op:
        # the same budget by opcode, priced with the measured issue cost of each instruction class
        scale = 250.0 / counts["spp"]
        ops, classes, by_phase = {}, {}, {}
        for row, c in rows.items():
            e, _ = execs(row)
            for op, n in c["ops"].items():
                cyc, cls = op_cycles(op)
                ops[op] = ops.get(op, 0.0) + n * e * scale
                classes[cls] = classes.get(cls, [0.0, 0.0])
                classes[cls][0] += n * e * scale
                classes[cls][1] += n * e * scale * cyc
                by_phase[row] = by_phase.get(row, [0.0, 0.0])
                by_phase[row][0] += n * e * scale
                by_phase[row][1] += n * e * scale * cyc
        tot_n = sum(v[0] for v in classes.values())
        tot_c = sum(v[1] for v in classes.values())
        print("VALU wave-instructions of the headline launch by issue class (static counts x wave-level executions; cost per instruction")
        print("measured with 6 waves per SIMD, scripts/ubench/valu_rate.hip -> profiles/r3/valu_issue_rates.txt)")
        for cls, (n, cy) in sorted(classes.items(), key=lambda kv: -kv[1][1]):
            print("  %-92s %10.4g inst %5.1f%%   x %.2f cycles = %5.1f%% of the issue cycles" % (cls, n, 100 * n / tot_n, cy / n, 100 * cy / tot_c))
        print("  total %.4g instructions, %.4g issue cycles = %.2f ms on 1024 SIMDs at 2.4 GHz (mean %.2f cycles per instruction)" % (
            tot_n, tot_c, tot_c / 1024 / 2.4e9 * 1e3, tot_c / tot_n))
        print("by phase (issue cycles):")
        for row, (n, cy) in sorted(by_phase.items(), key=lambda kv: -kv[1][1])[:14]:
            print("  %-82s %10.4g inst  %5.1f%% of the cycles  (%.2f cycles per instruction)" % (row, n, 100 * cy / tot_c, cy / n))
        print("the 25 most issued opcodes:")
        for op, n in sorted(ops.items(), key=lambda kv: -kv[1])[:25]:
            print("  %-28s %10.4g  %4.1f%%  %.2f cycles" % (op, n, 100 * n / tot_n, op_cycles(op)[0]))
        return
    total = 0.0
    out = []
    for row, c in rows.items():
        e, lanes = execs(row)
        dyn = c["valu"] * e
        total += dyn
        out.append((dyn, row, c, e, lanes / (64.0 * e) if e else 0.0))
    out.sort(reverse=True)
    scale = 250.0 / counts["spp"]
    print("Instruction budget of strict k_render<false,false>, headline frame (1280x720, %d spp counted%s), S = %d spheres" % (
        counts["spp"], ", scaled to 250 spp" if scale != 1.0 else "", S))
    print("wave-iterations %.4g, waves %d; rows: static instructions of the phase x wave-level executions" % (it * scale, waves))
    print("%-82s %6s %6s %5s %5s %5s %12s %7s %11s %6s" % ("phase", "VALU", "SALU", "SMEM", "LDS", "VMEM", "executions", "lanes", "VALU issued", "share"))
    for dyn, row, c, e, lanes in out:
        print("%-82s %6d %6d %5d %5d %5d %12.4g %6.1f%% %11.4g %5.1f%%" % (row, c["valu"], c["salu"], c["smem"], c["lds"], c["vmem"], e * scale, 100 * lanes,
                                                                              dyn * scale, 100 * dyn / total))
    print("%-82s %6d %51s %11.4g" % ("total (static VALU of the kernel | issued per launch)", sum(c["valu"] for c in rows.values()), "", total * scale))
    print("transcendental (v_rcp/v_sqrt/v_rsq) instructions per phase: " + ", ".join("%s: %d" % (r.split(":")[0] if ":" in r else r, c["trans"]) for _, r, c, _, _ in out if c["trans"]))
    lane_slots = sum(c["valu"] * lanes_ * e for _, _, c, e, lanes_ in out)
    print("active lanes per issued VALU instruction (weighted): %.1f%%" % (100.0 * lane_slots / total))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 250)
    elif len(sys.argv) > 1 and sys.argv[1] == "ops":
        table(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "phase_counts.json"), by_op=True)
    else:
        table(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "phase_counts.json"))
