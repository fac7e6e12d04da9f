for i in 1 2 3; do timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -1; done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
