cd /root/repo
for c in C3 C2 C4shard headline; do echo "== $c"; bash scripts/kstats.sh $c 2>&1 | head -22 | cut -c1-110; done
