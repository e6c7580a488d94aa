R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for f in 1 0; do
echo "LOCO_FUSE_CAT=$f $(LOCO_FUSE_CAT=$f python3 tests/diag/fwd_b1_time.py 1 100 2>&1 | grep 'B=') $(LOCO_FUSE_CAT=$f python3 tests/diag/decode_b25.py 2>&1 | grep 'B=')"
done; done
