export TMPDIR=/tmp
for i in 1 2 3; do
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
done
