#!/usr/bin/env python3
"""Issue / stall breakdown of a kernel from the counter passes of tools/pmc_issue.sh, read against the same counters on
tools/ubench/pmc_calib (streams whose vector-ALU load is known by construction).

    python tools/issue_breakdown.py gpurun_out/issue_r05_full [more dirs ...] > profiles/r05_full_issue_breakdown.txt

Units (rocprofv3 -L on gfx950, and the calibration below): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (4 clocks) summed
over waves; SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on every stream measured (one quad-cycle per instruction whatever it costs) and
SQ_THREAD_CYCLES_VALU equals 64 x SQ_INSTS_VALU -- neither says how busy the ALU is.  SQ_ACTIVE_INST_VALU2 counts quad-cycles in which a
SIMD issued TWO vector instructions, so
    quad-cycles of a SIMD in which it issued a vector instruction = (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / SIMDs
and, over the kernel's quad-cycles per SIMD (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 / 4), that is the share of time the vector
ALU's issue port is in use: 0.91-0.94 on the saturated calibration streams, 0.46 on a single dependent chain."""
import csv
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def load(d, prefix):
    acc = defaultdict(lambda: defaultdict(list))
    import glob
    for path in sorted(glob.glob("%s/%s*_counter_collection.csv" % (d, prefix))):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                k = k.split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v, skip=0):
    v = v[skip:] if len(v) > skip else v
    return sum(v) / len(v)


def report(name, c, skip, out):
    g = lambda k: mean(c[k], skip) if k in c else None                       # noqa: E731
    gui, insts, v2 = g("GRBM_GUI_ACTIVE"), g("SQ_INSTS_VALU"), g("SQ_ACTIVE_INST_VALU2")
    if not (gui and insts is not None and v2 is not None):
        return
    quads = gui / XCDS / 4.0
    issue_q = (insts - v2) / SIMDS
    out.append("%s" % name)
    out.append("  kernel: %.4g clocks per XCD = %.4g quad-cycles per SIMD;  vector instructions per SIMD %.4g (%.2f clocks each), %.1f %% of them issued two to a quad-cycle"
               % (gui / XCDS, quads, insts / SIMDS, gui / XCDS / (insts / SIMDS), 100 * 2 * v2 / insts))
    out.append("  ** %.1f %% of the SIMDs' quad-cycles issue a vector instruction **   (saturated calibration streams: 91-94 %%)" % (100 * issue_q / quads))
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        parts = [("issuing an instruction (SQ_ACTIVE_INST_ANY)", g("SQ_ACTIVE_INST_ANY")),
                 ("waiting to issue: pipe busy / dependency / arbitration (SQ_WAIT_INST_ANY)", g("SQ_WAIT_INST_ANY")),
                 ("   of which waiting on the LDS pipe (SQ_WAIT_INST_LDS)", g("SQ_WAIT_INST_LDS")),
                 ("parked in s_waitcnt / barrier: memory, LDS returns (SQ_WAIT_ANY)", g("SQ_WAIT_ANY"))]
        out.append("  per wave (of SQ_WAVE_CYCLES = %.4g quad-cycles; the three states below are disjoint and add up to it):" % wc)
        for label, v in parts:
            if v is not None:
                out.append("      %5.1f %%  %s" % (100 * v / wc, label))
        act = [(k[len("SQ_ACTIVE_INST_"):], g(k)) for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
                                                            "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_VMEM") if g(k) is not None]
        out.append("      issue slots by kind (quad-cycles / SQ_WAVE_CYCLES): " + ", ".join("%s %.1f %%" % (k, 100 * v / wc) for k, v in act))
    mix = [(k[len("SQ_INSTS_VALU_"):], g(k)) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_CVT",
                                                       "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_TRANS_F32") if g(k) is not None]
    if mix:
        rest = insts - sum(v for _, v in mix)
        out.append("  vector instruction mix: " + ", ".join("%s %.1f %%" % (k, 100 * v / insts) for k, v in mix) + ", other (moves, DPP, max, compares, perm) %.1f %%" % (100 * rest / insts))
    if g("SQ_INSTS_LDS"):
        out.append("  LDS: %.4g instructions per SIMD; bank conflicts %.1f %% of SQ_LDS_IDX_ACTIVE; data FIFO full %.3g, command FIFO full %.3g cycles; address conflicts %.3g"
                   % (g("SQ_INSTS_LDS") / SIMDS, 100 * (g("SQ_LDS_BANK_CONFLICT") or 0) / max(g("SQ_LDS_IDX_ACTIVE") or 1, 1), g("SQ_LDS_DATA_FIFO_FULL") or 0,
                      g("SQ_LDS_CMD_FIFO_FULL") or 0, g("SQ_LDS_ADDR_CONFLICT") or 0))
    if g("SQ_INSTS_VMEM_RD") is not None:
        out.append("  memory: %.4g loads + %.4g stores per SIMD, TA address FIFO full %.3g cycles; scalar %.4g, branches %.4g, instruction fetches %.4g per SIMD"
                   % (g("SQ_INSTS_VMEM_RD") / SIMDS, g("SQ_INSTS_VMEM_WR") / SIMDS, g("SQ_VMEM_TA_ADDR_FIFO_FULL") or 0, (g("SQ_INSTS_SALU") or 0) / SIMDS,
                      (g("SQ_INSTS_BRANCH") or 0) / SIMDS, (g("SQ_IFETCH") or 0) / SIMDS))
    out.append("")


def main():
    out = []
    for i, d in enumerate(sys.argv[1:]):
        out.append("=" * 8 + " " + d)
        if i == 0:
            cal = load(d, "calib")
            if cal:
                out.append("calibration: tools/ubench/pmc_calib under the same counter passes (second launch of each kind: 4000 x 64 instructions per wave)")
                names = {"calib<0>": "v_fma_f32, 8 independent chains, 4 waves / SIMD (saturated, fast class)",
                         "calib<1>": "v_cvt_f32_i32, 4 waves / SIMD (saturated, slow class)",
                         "calib<2>": "v_fma_f32 / v_cvt_f32_i32 alternating, 4 waves / SIMD (saturated)",
                         "calib<3>": "v_fma_f32, ONE dependent chain, 1 wave / SIMD (latency-bound)",
                         "calib<4>": "v_fma_f32, 8 chains, 1 wave / SIMD",
                         "calib<5>": "v_fma_f32 + 3 s_nop, 4 waves / SIMD"}
                for k in sorted(cal):
                    report(k + "  " + names.get(k, ""), {c: v[-1:] for c, v in cal[k].items()}, 0, out)
        for k, c in sorted(load(d, "p").items()):
            if "synth" in k or "checksum" in k or "rocclr" in k:
                continue
            report(k, c, 2, out)
    print("\n".join(out))


if __name__ == "__main__":
    main()
