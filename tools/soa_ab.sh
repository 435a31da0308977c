L=raytracinginaweekend_amd/lib
for i in 1 2; do
for lib in libmirt.so libmirt_soa.so; do
  echo "== $lib (lds_top=0)"
  MIRT_OPTS=lds_top=0 MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 128 spheres 3 | tail -1
  MIRT_OPTS=lds_top=0 MIRT_SIZE=1920x1080 MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 32 spheres1m 3 | tail -1
done; done
