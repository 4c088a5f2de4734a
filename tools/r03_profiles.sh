#!/bin/bash
# Round-3 rocprofv3 evidence for every BASELINE config (run on the GPU box from the repo root): kernel-trace + PMC passes of
# tools/profile.sh per workload; the summaries are copied to profiles/ by hand (gpurun_out/ is scratch).
set -u
for spec in "enc:--workload encode4096" "dec:--workload decode4096 --no-tolerant" "dectol:--workload decode4096 --tolerant" \
            "gray8k:--workload gray8k" "gray8k_dec:--workload gray8k_decode --no-tolerant" "batch1080p:--workload batch1080p" \
            "jpg:--workload encode4096_jpg"; do
  tag=${spec%%:*}; args=${spec#*:}
  tools/profile.sh r03_$tag $args > /dev/null 2>&1
  echo "== $tag"; grep -E "jpezy" gpurun_out/prof_r03_$tag/summary.txt | grep -E "calls=|FETCH_SIZE|WRITE_SIZE" | head -12
done
