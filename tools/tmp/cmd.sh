python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reset_all_lights_from_kept or occlusion_beside" 2>&1 | tail -15
