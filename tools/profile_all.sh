#!/bin/bash
# The round's profile set in one GPU-box visit: tools/profile_all.sh <round-tag>   (results under gpurun_out/prof_<tag>_<name>/)
R=${1:-r04}
bash tools/profile_round.sh ${R}_full full > /dev/null 2>&1
bash tools/profile_round.sh ${R}_full_two_kernels full --fused 0 --overlap 0 > /dev/null 2>&1
bash tools/profile_round.sh ${R}_wf wf > /dev/null 2>&1
bash tools/profile_round.sh ${R}_mixed mixed > /dev/null 2>&1                                   # ssdr_run_chain's default: the two stages side by side
bash tools/profile_round.sh ${R}_mixed_serial mixed --concurrent 2 --overlap 0 > /dev/null 2>&1  # every kernel on its own: per-kernel durations
bash tools/profile_round.sh ${R}_wf_hop512 wf --hop 512 > /dev/null 2>&1
bash tools/profile_round.sh ${R}_full_hop512 full --hop 512 > /dev/null 2>&1                     # default at hop 512: two kernels side by side
bash tools/profile_round.sh ${R}_full_hop512_fused full --hop 512 --fused 2 > /dev/null 2>&1     # the one-read kernel at hop 512 (opt-in)
bash tools/profile_round.sh ${R}_wf_exact wf --exact 1 > /dev/null 2>&1
bash tools/profile_round.sh ${R}_decim4 decim4 > /dev/null 2>&1                                  # the decimating front end (D = 4)
bash tools/profile_round.sh ${R}_am_narrow am_narrow > /dev/null 2>&1                            # all-AM at +-4 kHz, every channel on the general audio path: ssdr_run_chain's wave-specialised kernel
bash tools/profile_round.sh ${R}_am_narrow_two_kernels am_narrow --fused 0 > /dev/null 2>&1      # ... and the general audio kernel beside the waterfall kernel (round 5's default)
bash tools/profile_round.sh ${R}_mixed_chain_ws mixed --fused 3 > /dev/null 2>&1                 # the wave-specialised one-read kernel on configs[3] (opt-in there)
for d in gpurun_out/prof_${R}_*; do echo "== $d"; head -6 $d/kernel_stats.txt; done
