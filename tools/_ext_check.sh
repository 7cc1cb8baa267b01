cd ${GRAFT_REPO_ROOT:-/root/repo}
FGA_EXTEND_PROFILE=1 timeout 60 python bench.py --steps 3 --warmup 1 --no-cpu --no-cold 2>&1 | grep -E "extend profile|ms_per_step" | sed -e 's/.*kernel \([0-9.]* ms\).*/kernel \1/' | cut -c1-200 | tail -3
FGA_SKIP_1G=1 timeout 120 python -m pytest tests -m gpu -q -x -k "end_to_end or extend or golden or shims or config2 or divergent or s1_86" 2>&1 | tail -2
