#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_more.py tests/test_gpu_drafter.py tests/test_drafter_layer.py tests/test_gpu_loop.py -m gpu -q -k "head_sample or static_plan or additive or raw_row_throughput or tree_attention_path or static_tree_v1 or lumina_topK_generate or static_loop_calls" > $O/test.log 2>&1; echo "tests rc=$?"; tail -40 $O/test.log
