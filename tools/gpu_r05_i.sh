#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python -c "import __graft_entry__ as g; g.build_tools()" 
for i in 1 2; do
echo "default loads:"; timeout 300 python tools/fb_wgrad_fusion_probe.py 2>&1 | grep "as built"
echo "non-temporal loads:"; CDA_TOOLS_LIB=$R/tools/libcda_tools_nt.so timeout 300 python tools/fb_wgrad_fusion_probe.py 2>&1 | grep "as built"
done
