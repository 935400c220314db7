#!/bin/bash
# round 4, GPU call 1 (diagnosis): MFMA/VALU overlap micro-benchmark, attention variants A/B, isolated vs in-situ encoder launches
# (plain, then under rocprofv3 with GRBM_GUI_ACTIVE and FETCH_SIZE), a short baseline bench line for this box.
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c1 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 120 tools/ubench/overlap > $O/overlap.txt 2>&1
timeout 600 python tools/r4_attn_ab.py --rounds 2 base noprio nomax nomax_noprio st3 st3_nomax w8st4 w8st4_nomax w8st2_nomax oagpr > $O/attn_ab.txt 2>&1
timeout 400 python tools/r4_insitu.py > $O/insitu.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_clk -o c -- python $R/tools/r4_insitu.py --quick --phases-json $R/$O/phases_clk.json > $R/$O/insitu_clk.log 2>&1
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/tools/r4_insitu.py --quick --phases-json $R/$O/phases_fetch.json > $R/$O/insitu_fetch.log 2>&1 )
python tools/r4_insitu_post.py $O/pmc_clk $O/phases_clk.json > $O/insitu_clk.txt 2>&1
python tools/r4_insitu_post.py $O/pmc_fetch $O/phases_fetch.json > $O/insitu_fetch.txt 2>&1
rm -rf $O/pmc_clk $O/pmc_fetch            # raw traces are large; the sliced tables are what is kept
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.txt 2> $O/bench.err
cat $O/overlap.txt; cat $O/attn_ab.txt; cat $O/insitu.txt | grep -v JSON; echo CLK; cat $O/insitu_clk.txt | cut -c1-220; echo FETCH; cat $O/insitu_fetch.txt | cut -c1-220; head -c 700 $O/bench.txt
