# final evidence of the round: bench lines of the four configurations (+ the Python-route and two-ranks-one-card lines), then
# rocprofv3 passes for the two configurations whose kernels changed last (K7, K9)
bash scripts/gpu_bench_lines_r3.sh 2>&1 | grep -v "^r3_k9\|prof_r3_k9" | tail -12
PROFILE_TAGS="k7" bash scripts/gpu_profiles_r3.sh
