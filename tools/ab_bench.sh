# same-box A/B: HEAD's tree (a git worktree built under _ab_head/: `git worktree add -f _ab_head HEAD && (cd _ab_head && python __graft_entry__.py)`) against this tree,
# alternating; tools/ab_report.py prints the means.  Experimental builds of this tree's library: tools/build_variant.sh + CDA_HIP_LIB.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python _ab_head/bench.py --no-cpu-baseline --no-league-leg > gpurun_out/ab_head_1000_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-league-leg > gpurun_out/ab_new_1000_$i.json 2>/dev/null
done
for i in 1 2 3; do
python _ab_head/bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_head_20_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_new_20_$i.json 2>/dev/null
done

