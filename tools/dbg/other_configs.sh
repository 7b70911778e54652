run() { echo "== $*"; timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --cpu-cif-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'MB/s', round(d['ms_per_step'],1), 'ms/step BER', d['extracted_payload_BER'], d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],2))"; }
run --width 1280 --height 720 --me hex --subme 6 --gops 4096
run --width 3840 --height 2160 --me esa --subme 6 --gops 512
run --width 352 --height 288 --me dia --subme 6 --emrate 35 --gops 8192
run --width 1280 --height 720 --me hex --subme 5 --gops 512
run --width 3840 --height 2160 --me esa --subme 5 --gops 16
run --width 352 --height 288 --me dia --subme 5 --emrate 35 --gops 2048
