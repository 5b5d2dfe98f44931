#!/bin/bash
# __graft_entry__.smoke() on a GPU box: one tiny bank build + evaluation through the C ABI, checked against the oracle
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
