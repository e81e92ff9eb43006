# build and run the MFMA issue-rate micro-benchmark on the GPU box: bash tools/micro/run.sh
set -e
here="$(cd "$(dirname "$0")" && pwd)"
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_issue "$here/mfma_issue.hip"
timeout -k 10 60 /tmp/mfma_issue
