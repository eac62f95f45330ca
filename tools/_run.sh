mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','vit_forward_ms','vit_forward_frac_of_bf16_peak')})"
