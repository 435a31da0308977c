set -e
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python -u tools/quickbench.py 1024 cornell 3 2>&1| tail -2
