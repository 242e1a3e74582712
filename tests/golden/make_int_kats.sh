#!/bin/sh
# Regenerates tests/golden/int_kats.json from the reference's own hash.h / pcg32.h.
# Only works where /root/reference exists (this container); the JSON it writes is the
# committed fixture that travels to the GPU box.
set -e
cd "$(dirname "$0")/../../oracle"
make ref
./_ref/kat_ref > ../tests/golden/int_kats.json
echo "wrote tests/golden/int_kats.json"
