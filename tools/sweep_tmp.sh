set -e
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -5
export MIRT_DEBUG=1
for e in 1 0; do echo "== cornell engine=$e";  MIRT_OPTS=engine=$e timeout -k 10 200 python -u tools/quickbench.py 1024 cornell 2 2>&1| tail -2; done
for e in 1 0; do echo "== 1m engine=$e";  MIRT_OPTS=engine=$e timeout -k 10 200 python -u tools/quickbench.py 64 spheres1m 2 2>&1| tail -2; done
for e in 1 0; do echo "== spheres engine=$e";  MIRT_OPTS=engine=$e timeout -k 10 200 python -u tools/quickbench.py 64 spheres 2 2>&1| tail -1; done
for e in 1 0; do MIRT_OPTS=engine=$e timeout -k 10 200 python -u tools/quickbench.py 4 cornell 2 2>&1| tail -1; done
