export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "multi or gather" 2>&1 | tail -15
