cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "dp or extras or unet" 2>&1 | grep -v "^  File\|Extension modules" | tail -5
