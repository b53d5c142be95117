#!/bin/bash
# development helper: env-step roofline of several library builds, alternating, same box
for rep in 1 2; do for lib in "$@"; do echo "$lib"; SES_LIB_PATH=$PWD/$lib python tools/roofline_variance.py | head -3 | tr '\n' ' '; echo; done; done
