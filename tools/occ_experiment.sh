for cfg in "1 0" "1 42000" "1 70000" "4 0" "4 13000" "4 21000" "4 33000"; do set -- $cfg; echo "variant=$1 pad=$2"; VS_MICRO_REPS=2 VS_CONV_VARIANT=$1 VS_CONV_LDS_PAD=$2 timeout 300 python tools/conv_micro.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:v['tflops'] for k,v in d.items() if isinstance(v,dict) and 'mish' in k})"; done
