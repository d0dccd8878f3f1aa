# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# do the stage ranges reach a rocprofv3 --marker-trace summary?  (GPU box)
cd /tmp && export TMPDIR=/tmp
for mode in auto forced; do
  rm -rf /tmp/rx_$mode
  if [ $mode = forced ]; then export MI_ROCTX=1; fi
  rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/rx_$mode -o r -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1
  echo "== $mode"; ls /tmp/rx_$mode | head; f=$(find /tmp/rx_$mode -name "*marker_api_stats.csv" | head -1); [ -n "$f" ] && cat $f | head -12 | cut -c1-160
done
