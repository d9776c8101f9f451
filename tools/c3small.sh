CONFIG3_SMALL=1 python - <<'PY'
import sys; sys.path[:0]=['tests']
import test_config3_gpu as t
t.test_config3_exact_partition_eight_ranks_on_one_gpu(64)
print("small ok")
PY
