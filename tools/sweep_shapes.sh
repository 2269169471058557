SHAPES=("4 4" "6 3" "8 2" "5 3" "6 2" "5 4")
# usage: bash tools/sweep_shapes.sh   (threads x depth shapes of the decode pool, one pipelined bench line each)
for shape in "${SHAPES[@]:-4 2}"; do set -- $shape
python bench.py --no-tunstall-scaled --no-cpu --no-other-configs --host-threads $1 --depth $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$shape', d['value'], d['ms_per_step'], d['steady_state'])"
done
