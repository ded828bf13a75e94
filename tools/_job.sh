cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== sleep 8 (default)"; timeout 120 tools/_bin/chol_test 6016 10 | grep "n= 6016"
for sl in 1 2 4 16 32; do echo "== sleep $sl"; timeout 120 tools/_bin/chol_test_s$sl 6016 10 | grep "n= 6016"; done
done
