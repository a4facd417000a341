#!/usr/bin/env python3
"""profiles/rNN_isa_histograms.txt: opcode histograms of the kernels' steady-state loops at HEAD (static counts from hipcc -save-temps)
and the FMA share bench.py's issue-slot figure uses.   python tools/isa_round.py > profiles/r04_isa_histograms.txt"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "supersdr_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function".split()
KERNELS = [("ssdr_wf.hip", "ssdr_wf_kernelILb0ELb0E", "ssdr_wf_kernel<false, false>"),
           ("ssdr_wf.hip", "ssdr_wf_kernelILb1ELb0E", "ssdr_wf_kernel<true, false>"),
           ("ssdr_wf.hip", "ssdr_wf_kernelILb0ELb1E", "ssdr_wf_kernel<false, true>"),
           ("ssdr_wf.hip", "ssdr_fused_am_kernelILb0E", "ssdr_fused_am_kernel<false>  (hop 1024)"),
           ("ssdr_wf.hip", "ssdr_fused_am_kernelILb1E", "ssdr_fused_am_kernel<true>   (hop 512)"),
           ("ssdr_audio.hip", "ssdr_audio_kernelILi0E", "ssdr_audio_kernel<0>  general: NCO -> FIR -> demodulator"),
           ("ssdr_audio.hip", "ssdr_audio_kernelILi1E", "ssdr_audio_kernel<1>  full-band lane shift"),
           ("ssdr_audio.hip", "ssdr_audio_kernelILi2E", "ssdr_audio_kernel<2>  full-band AM (no NCO, no FIR)"),
           ("ssdr_audio.hip", "ssdr_audio_dec_kernelILi4E", "ssdr_audio_dec_kernel<4>"),
           ("ssdr_wf_exact.hip", "ssdr_wf_exact_kernelILb0ELb0ELb0E", "ssdr_wf_exact_kernel<false, false>  (float64)"),
           ("ssdr_post.hip", "ssdr_db2col_kernel", "ssdr_db2col_kernel"),
           ("ssdr_post.hip", "ssdr_play_kernel", "ssdr_play_kernel"),
           ("ssdr_post.hip", "ssdr_play_rs_kernel", "ssdr_play_rs_kernel"),
           ("ssdr_post.hip", "ssdr_iqwire_kernel", "ssdr_iqwire_kernel")]
FMA = re.compile(r"^v_(pk_)?(fma|fmac|fmaak|fmamk|mad)_(f32|f64)")


def body(s, name):
    m = re.search(r'^(\S*%s\S*):.*?\n(.*?)\n\s*s_endpgm' % re.escape(name), s, re.S | re.M)
    return m.group(2).split('\n') if m else None


def main():
    print(__doc__.strip().splitlines()[0])
    print("Static counts: a loop holds the code of every branch it can take, so totals exceed what one unit executes; the dynamic counts are the PMC\n"
          "SQ_INSTS_VALU figures in the rNN_*_pmc_summary.txt files.  FMA share = fused multiply-add opcodes / VALU opcodes of the loop.\n")
    tmp = tempfile.mkdtemp()
    asm = {}
    for f in sorted({k[0] for k in KERNELS}):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-save-temps=obj", "-c", os.path.join(CS, f), "-o", os.path.join(tmp, f + ".o")],
                              cwd=CS, stderr=subprocess.DEVNULL)
        asm[f] = open(os.path.join(tmp, f.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for f, sym, label in KERNELS:
        lines = body(asm[f], sym)
        if lines is None:
            print("== %s: not found" % label)
            continue
        lab = {}
        for i, l in enumerate(lines):
            mm = re.match(r'^(\.LBB\d+_\d+):', l)
            if mm:
                lab[mm.group(1)] = i
        best = (0, len(lines) - 1)
        span = 0
        for i, l in enumerate(lines):
            mm = re.match(r'\s+s_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in lab and lab[mm.group(1)] < i and i - lab[mm.group(1)] > span:
                best, span = (lab[mm.group(1)], i), i - lab[mm.group(1)]
        c = collections.Counter()
        for l in lines[best[0]:best[1] + 1]:
            t = l.strip()
            if not t or t[0] in ';.' or t.endswith(':'):
                continue
            c[t.split()[0]] += 1
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        fma = sum(v for k, v in c.items() if FMA.match(k))
        lds = sum(v for k, v in c.items() if k.startswith('ds_'))
        vmem = sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))
        print("== %s\n   loop lines %d..%d: %d instructions, %d VALU (FMA share %.2f), %d LDS, %d VMEM"
              % (label, best[0], best[1], sum(c.values()), valu, fma / max(valu, 1), lds, vmem))
        for k, v in c.most_common(18):
            print("     %-34s %d" % (k, v))
        print()


if __name__ == "__main__":
    main()
