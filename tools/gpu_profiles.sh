#!/bin/bash
# GPU-box session: the rocprofv3 evidence (kernel trace + PMC passes) of the three measured forward configurations;
# tools/collect_profiles.py then copies the summaries into profiles/.
set -u
bash tools/profile.sh b16_224_bs256 b16 224 256 || exit $?
bash tools/profile.sh l16_224_bs256 l16 224 256 || exit $?
bash tools/profile.sh b16_384_bs128 b16 384 128 || exit $?
