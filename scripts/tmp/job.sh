cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m pytest tests -m gpu -q 2>&1 | tail -3
