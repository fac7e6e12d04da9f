export TMPDIR=/tmp
bash tools/profile_round.sh r02 cfg4 2>&1 | tail -12
bash tools/profile_round.sh r02 cfg3 2>&1 | tail -8
bash tools/profile_round.sh r02 cfg2 2>&1 | tail -8
ls gpurun_out/profiles
