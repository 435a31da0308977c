set -e
for o in "max_paths=67108864" "max_paths=268435456" "max_paths=536870912" "max_paths=1073741824"; do echo "== $o";  MIRT_OPTS=$o timeout -k 10 200 python -u tools/quickbench.py 1024 cornell 2 2>&1| tail -1; done
