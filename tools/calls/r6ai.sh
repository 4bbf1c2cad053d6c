#!/bin/bash
# round 6, call 36: the tests that touch split-K linears, on the final libraries
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "long_k_split or test_linear" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu -k "simplecnn or SimpleCNN or config3 or probe or low_level or golden or parity" 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
