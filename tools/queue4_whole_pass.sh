B="python3 bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 16"
run() { printf "Q=%s own=%s pad=%s : " "$1" "$2" "$3"; GPU_MAX_HW_QUEUES=$1 WIPA_OWN_STREAMS=$2 WIPA_PAD_STREAMS=$3 timeout -k 10 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms per pass', d['value'],'audio-s/s | hw_queues', d['config']['hw_queues'],'| evaluate-style', d['evaluate_style']['frac_of_value'])" || exit 1; }
run 8 0 0; run 4 1 0; run 8 1 0; run 4 1 0; run 8 0 0
