#!/bin/bash
# A/B of the working library against variants/lib_prev.so on one box: C2 (narrow), top-100, C3 shape (wide, 50k queries).
tag=${1:-r03}
bash scripts/ab_multi.sh $tag variants/lib_prev.so
