for cfg in "3 2" "4 2" "3 3" "4 3" "2 4" "3 2" "4 2" "3 3"; do set -- $cfg
 timeout 120 python bench.py --steps 480 --warmup 48 --no-cpu --no-other-configs --depth $1 --host-threads $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $1 threads $2', d['value'], d['ms_per_step'], d['bit_exact'])"
done
