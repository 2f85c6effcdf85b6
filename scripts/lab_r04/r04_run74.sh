cd /root/repo
for c in C2 C3 C4shard; do echo "== $c"; scripts/kstats.sh $c 2>&1 | cut -c1-200 | sed 's/"host_ms.*//'; done
