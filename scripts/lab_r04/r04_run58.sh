cd /root/repo
TAG=xm bash scripts/kstats.sh headline --exact-masks 2>&1 | head -14 | cut -c1-120
