# round-3 rocprofv3 evidence for the four BASELINE configurations (scripts/profile.sh: one trace pass + six PMC passes each)
PROFILE_STEPS=20 PROFILE_WARMUP=10 bash scripts/profile.sh r3_k7_default --config 1
PROFILE_STEPS=20 PROFILE_WARMUP=10 bash scripts/profile.sh r3_hard8 --config 3
PROFILE_STEPS=10 PROFILE_WARMUP=4 bash scripts/profile.sh r3_k9 --config 2
PROFILE_STEPS=5 PROFILE_WARMUP=2 bash scripts/profile.sh r3_k15 --config 4
