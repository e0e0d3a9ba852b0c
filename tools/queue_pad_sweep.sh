run() { printf "Q=%-2s own=%s pad=%s batch=%-2s : " "$1" "$2" "$3" "$4"; GPU_MAX_HW_QUEUES=$1 WIPA_OWN_STREAMS=$2 WIPA_PAD_STREAMS=$3 timeout -k 10 200 python bench.py --phase dec --batch $4 --pipeline 4 --steps 8 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_pass'],'ms per pass')" || exit 1; }
run 8 0 0 8
for Q in 4 8; do for PAD in 0 1 2 3 4 7; do run $Q 1 $PAD 8; done; done
