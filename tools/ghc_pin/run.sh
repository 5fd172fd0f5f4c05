#!/bin/sh
# Runs the pinning kit end to end wherever `stack` (resolver lts-13.16, GHC 8.6) works -- NOT in the image this repository is built in.
#   tools/ghc_pin/run.sh /path/to/checkout/of/flannelhead-blackstar
# Copies Dump.hs + inputs into the checkout, appends the executable stanza to its blackstar.cabal (once), builds, runs both input sets
# and puts the dumps where this repository's tests look for them (tests/golden/ghc/).  Nothing in the checkout's src/ is touched.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
REF=${1:?usage: run.sh /path/to/blackstar-checkout}
command -v stack >/dev/null || { echo "stack not found: this needs GHC 8.6 / stack with resolver lts-13.16" >&2; exit 1; }
mkdir -p "$REF/tools/ghc_pin"
cp "$HERE/Dump.hs" "$REF/tools/ghc_pin/Dump.hs"
rm -rf "$REF/tools/ghc_pin/inputs" && cp -r "$HERE/inputs" "$REF/tools/ghc_pin/inputs"
if ! grep -q '^executable ghc-pin-dump' "$REF/blackstar.cabal"; then
cat >> "$REF/blackstar.cabal" <<'CABAL'

executable ghc-pin-dump
  hs-source-dirs:      tools/ghc_pin
  main-is:             Dump.hs
  ghc-options:         -Wall -O2 -rtsopts -threaded -with-rtsopts=-N
  build-depends:       base, blackstar, bytestring, cereal, yaml, kdt, linear, massiv, massiv-io, directory, filepath
  default-language:    Haskell2010
CABAL
fi
(cd "$REF" && stack build)
for set in uniform clustered; do
  rm -rf "$REF/tools/ghc_pin/out/$set"
  (cd "$REF" && stack exec ghc-pin-dump -- "tools/ghc_pin/inputs/$set" "tools/ghc_pin/out/$set")
  mkdir -p "$REPO/tests/golden/ghc"
  rm -rf "$REPO/tests/golden/ghc/$set" && cp -r "$REF/tools/ghc_pin/out/$set" "$REPO/tests/golden/ghc/$set"
done
# the FFI shim of INTEGRATION.md section 1 and the batch-mode edit of doStart (BatchMain imports RaytracerFFI, so one command checks
# both): type-checked against the reference's own modules (no GPU, no library needed for that)
cp "$HERE/RaytracerFFI.hs" "$HERE/BatchMain.hs" "$REF/tools/ghc_pin/"
if (cd "$REF" && stack ghc -- -fno-code -Wall -isrc -itools/ghc_pin tools/ghc_pin/RaytracerFFI.hs tools/ghc_pin/BatchMain.hs) > "$REPO/tests/golden/ghc/shim_typecheck.log" 2>&1; then
  echo "OK" > "$REPO/tests/golden/ghc/shim_typecheck.txt"
else
  echo "FAILED (see shim_typecheck.log)" > "$REPO/tests/golden/ghc/shim_typecheck.txt"
fi
echo "dumps are in $REPO/tests/golden/ghc/ -- now: python -m pytest tests/test_oracle.py -k reference_itself -rs"
