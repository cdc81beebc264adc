cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in v0 v1 v2 v3 v4 v5 v6 v7; do echo "== $v"; FSAMD_LIB=exp/lib/libfs_$v.so timeout -k 10 100 python -u scripts/time_pairs.py 256 512 | grep "one sweep per pass"; done
for v in v0 a1 a3; do echo "== $v advect"; FSAMD_LIB=exp/lib/libfs_$v.so timeout -k 10 100 python -u scripts/time_advect.py 512 2>&1 | tail -3; done
