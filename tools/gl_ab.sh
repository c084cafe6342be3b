run() { name=$1; shift; for i in 1 2; do env "$@" timeout 200 python bench.py --workload goldilocks_lde_2p20 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/gl_$name.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/gl_$name.json')); print('$name', round(d['value']*1e3,3), 'ms', d['roofline']['launches'], 'launches', round(d['roofline']['streamed_bytes_per_s']/1e12,2), 'TB/s streamed')"; done; }
run default A=1
run t14r3w512 SS_GL_LOG_TILE=14 SS_GL_MIN_RUN_LOG=3 SS_GL_THREADS=512
run t14r3w256 SS_GL_LOG_TILE=14 SS_GL_MIN_RUN_LOG=3 SS_GL_THREADS=256
run t14r5w512 SS_GL_LOG_TILE=14 SS_GL_MIN_RUN_LOG=5 SS_GL_THREADS=512
run t13r5w512 SS_GL_THREADS=512
SS_GL_LOG_TILE=14 SS_GL_MIN_RUN_LOG=3 SS_GL_THREADS=512 timeout 400 python -m pytest tests/test_goldilocks.py -m gpu -q -x 2>&1 | tail -2
