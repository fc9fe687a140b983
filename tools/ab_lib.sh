# Shared by the A/B and variant scripts (source it): a build whose failure is NOT silent -- a variant that does not
# compile must never be timed as the previously built library under its label -- and a backup of the
# file under test that is restored on every way out, interrupted runs included.
ab_build() {   # [extra hipcc flags]   -> returns non-zero (and says so) when the build fails
  local log
  log=$(SC_EXTRA_HIPCC_FLAGS="${1:-}" python springcraft_amd/csrc/build.py 2>&1)
  local rc=$?
  if [ $rc -ne 0 ]; then
    echo "BUILD FAILED (${1:-no extra flags}): variant skipped" >&2
    echo "$log" | tail -5 >&2
  fi
  return $rc
}
ab_keep() {    # <file>: keep a copy in a private temp file, restore it (and rebuild) on EXIT
  AB_FILE=$1
  AB_BACKUP=$(mktemp /tmp/ab_keep.XXXXXX) || exit 1
  cp "$AB_FILE" "$AB_BACKUP"
  trap 'cp "$AB_BACKUP" "$AB_FILE"; rm -f "$AB_BACKUP"; touch "$AB_FILE"; ab_build "" > /dev/null' EXIT
}
