#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_bench_contract.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | cut -c1-400
