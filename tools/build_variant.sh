#!/bin/bash
# Build an A/B variant of the library WITHOUT touching the shipped sources: copy supersdr_amd/csrc to a scratch tree, apply the named
# experiment patches (tools/experiments/*.patch), compile with the given -D flags into supersdr_amd/libssdr_<name>.so
# (git-ignored; travels to the GPU box; selected with SSDR_LIB_PATH, see tools/ab_bench.sh).
#   tools/build_variant.sh <name> "<-D flags>" [patch ...]
#   tools/build_variant.sh mfma   "-DSSDR_FIR_MFMA=1"      ssdr_audio_switches
#   tools/build_variant.sh abl1   "-DSSDR_FUSED_ABLATE=1"  ssdr_wf_switches
# The experiment switches the patches restore (each measured and recorded in profiles/HISTORY.md; none ships):
#   ssdr_audio_switches: SSDR_FIR_MFMA (channel FIR on the f32 MFMA), SSDR_AUDIO_PREFETCH (the stand-alone audio kernels; the wave-specialised kernel prefetches as shipped)
#   ssdr_chain_ws_hop512: the wave-specialised kernel at hop 512 (profiles/r06_ab_chain_ws.txt: 0 to -3.7 %)
#   ssdr_chain_ws_knobs: that kernel's tuning constants as -DSSDR_WS_... switches (ring depth, prefetch, poll naps, priorities, workgroup shape)
#   ssdr_wf_switches:    SSDR_WF_ABLATE, SSDR_FUSED_ABLATE (timing ablations), SSDR_WF_PAIR_MAJOR, SSDR_WF_BLOCKED_ITEMS, SSDR_FUSED_WIDE_LOADS=0
set -e
NAME=$1; FLAGS=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/supersdr_amd $T/include
cp -r $ROOT/supersdr_amd/csrc $T/supersdr_amd/csrc
cp $ROOT/include/ssdr.h $T/include/
for p in "$@"; do
  patch -s -p1 -d $T < $ROOT/tools/experiments/$p.patch
done
make -s -C $T/supersdr_amd/csrc OUT=$ROOT/supersdr_amd/libssdr_$NAME.so EXTRA="$FLAGS"
rm -rf $T
echo "built supersdr_amd/libssdr_$NAME.so ($FLAGS; patches: $*)"
