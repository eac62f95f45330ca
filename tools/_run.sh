mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_dp_sim_gpu.py -m gpu -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "fp32 or output_fields" 2>&1 | grep -v "OK$" | grep "fp32 mode:\|passed\|failed\|Error\|error" | head -20
