#!/bin/bash
# Build the stamped diagnostic variant of the c3 inverse kernel into ablate_build/libfinc_stamp.so (never shipped).
exec "$(dirname "$0")/build_variant.sh" stamp -DFINC_STAMP "$@"
