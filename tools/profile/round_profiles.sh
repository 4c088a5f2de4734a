#!/bin/bash
# rocprofv3 evidence of a round for every BASELINE config (run on the GPU box from the repo root): kernel-trace + PMC passes of
# tools/profile/profile.sh per workload, written under gpurun_out/prof_<round>_<tag>/; the summaries are copied to profiles/ afterwards
# (gpurun_out/ is scratch).   usage: tools/profile/round_profiles.sh r04
set -u
R=${1:-r04}
for spec in "enc:--workload encode4096" "dec:--workload decode4096 --no-tolerant" "dectol:--workload decode4096 --tolerant" \
            "gray8k:--workload gray8k" "gray8k_dec:--workload gray8k_decode --no-tolerant" "batch1080p:--workload batch1080p" \
            "jpg:--workload encode4096_jpg" "jpgdec:--workload decode4096_jpg"; do
  tag=${spec%%:*}; args=${spec#*:}
  tools/profile/profile.sh ${R}_$tag $args > /dev/null 2>&1
  echo "== $tag"; grep -E "jpezy" gpurun_out/prof_${R}_$tag/summary.txt | grep -E "calls=|FETCH_SIZE|WRITE_SIZE" | head -12
done
