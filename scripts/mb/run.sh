#!/bin/bash
set -e
cd "$(dirname "$0")"
for f in "$@"; do hipcc --offload-arch=gfx950 -O3 -o /tmp/$f $f.hip && /tmp/$f; done
