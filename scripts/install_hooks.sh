#!/bin/bash
# Installs a post-commit hook that stamps the new commit into .git_head (git-ignored; it travels with the working tree to
# the GPU box, which gets no .git): bench.py's `git_head` field then names the commit the snapshot was taken from even when
# somebody else (the round driver) takes the snapshot.   usage: bash scripts/install_hooks.sh
R=$(cd $(dirname $0)/.. && pwd)
H=$R/.git/hooks/post-commit
mkdir -p $R/.git/hooks
cat > $H <<'HOOK'
#!/bin/sh
git rev-parse HEAD > "$(git rev-parse --show-toplevel)/.git_head" 2>/dev/null || true
HOOK
chmod +x $H
git -C $R rev-parse HEAD > $R/.git_head
echo "installed $H; .git_head = $(cat $R/.git_head)"
