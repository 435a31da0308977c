cd $GRAFT_REPO_ROOT
L=raytracinginaweekend_amd/lib
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_textures.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4
for i in 1 2; do
for lib in libmirt_base.so libmirt.so; do
echo "== $lib"
MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 512 cornell 3 | tail -1
MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 128 spheres 3 | tail -1
done; done
