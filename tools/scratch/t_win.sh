cd $GRAFT_REPO_ROOT
for w in 320 400 480; do
for a in "0.47 0.45" "0.7 0.6"; do
echo "window $w args $a"
PGL_FLIP_WINDOW=$w python tools/probe_flipweights.py $a 2 2>&1 | tail -1 | cut -c1-330
done; done
