L=raytracinginaweekend_amd/lib
for i in 1 2 3; do for lib in libmirt_base.so libmirt.so; do echo -n "$lib  "; MIRT_LIB=$L/$lib python tools/quickbench.py 512 cornell 3 | tail -1; done; done
for lib in libmirt_base.so libmirt.so; do echo -n "$lib  "; MIRT_LIB=$L/$lib python tools/quickbench.py 100 spheres4 3 | tail -1; done
