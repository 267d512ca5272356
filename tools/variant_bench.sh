for v in "$@"; do echo "== $v"; DINT_HIP_LIB=$PWD/dint_amd/variants/$v.so python bench.py --cpu-seconds 0 --no-verify 2>/dev/null | python tools/bench_line.py; done
