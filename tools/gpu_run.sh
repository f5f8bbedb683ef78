#!/bin/bash
# The GPU-side steps of a development round as ONE parameterised script (replaces the one-shot tools/r04/run_*.sh
# files of round 4).  Run from the repo root on the GPU box, i.e. through gpurun:
#
#     gpurun --timeout 3000 -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
#
# Every step writes under gpurun_out/<tag>/ and appends one line to gpurun_out/<tag>/summary.txt.  Steps:
#   suite            the whole GPU suite (no -x: every failure is listed)
#   suite:<expr>     pytest -k <expr>
#   poison:<mode>    the suite with SSA_POISON=<nan|big> (uninitialised device buffers filled, tools/poison.py)
#   smoke            __graft_entry__.smoke()
#   bench            python bench.py --steps 20 --warmup 5 (what the driver runs)
#   profiles:<tag>   tools/collect_profiles.py <tag> (rocprofv3 kernel trace + PMC passes of bench.py)
#   hunt:<reps>      tools/chol_race_hunt.py <reps>: cold factorizations of the 4-film stack, bit-compared
#   repeat:<reps>    tools/config5_repeat.py <reps>: config 5 in both self-field modes, bit-compared
#   qform            tools/q_form_timing.py: Q assembly, one-shot pieces against strips
#   timeline:<kind>  rocprofv3 kernel trace of tools/fact_timeline.py run <kind> (float64|float32|stack4|single), then its
#                    `table` (SYRK launches in situ / alone / beside) and `rounds` listings
#   passes           rocprofv3 kernel trace of three warm solves of config H (tools/pass_trace.py) and its per-kernel listing
#   ab:<args>        tools/ab_knobs.py <args> (SSA_CHOL_DEBUG variants taking turns in one process)
#   py:<script and args>   any tool, e.g. "py:tools/fact_single.py 129"
#   env:<NAME=VALUE>       exported for the steps that follow (e.g. env:SSA_CHOL_DEBUG=late=1)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
sum=$out/summary.txt
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    env)      export "$arg"; echo "env $arg" >> $sum ;;
    suite)    if [ -n "$arg" ]; then sel=(-k "$arg"); else sel=(); fi
              timeout 1800 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 "${sel[@]}" > $out/pytest_gpu.log 2>&1
              echo "suite[$arg] rc=$? $(tail -1 $out/pytest_gpu.log)" >> $sum ;;
    poison)   SSA_POISON=$arg timeout 1800 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu_$arg.log 2>&1
              echo "suite under SSA_POISON=$arg rc=$? $(tail -1 $out/pytest_gpu_$arg.log)" >> $sum ;;
    smoke)    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
              echo "smoke rc=$? $(tail -1 $out/smoke.log)" >> $sum ;;
    bench)    timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
              echo "bench rc=$?" >> $sum ;;
    profiles) timeout 2400 python tools/collect_profiles.py $arg $out > $out/collect.log 2>&1
              echo "profiles[$arg] rc=$?" >> $sum ;;
    hunt)     timeout 1200 python -X faulthandler tools/chol_race_hunt.py $arg > $out/hunt.txt 2>&1
              echo "hunt rc=$? $(tail -1 $out/hunt.txt)" >> $sum ;;
    repeat)   timeout 900 python -X faulthandler tools/config5_repeat.py $arg > $out/config5_repeat.txt 2>&1
              echo "repeat rc=$? $(tail -1 $out/config5_repeat.txt)" >> $sum ;;
    qform)    timeout 900 python -X faulthandler tools/q_form_timing.py $arg > $out/q_form_timing.txt 2>&1
              echo "qform rc=$?" >> $sum ;;
    timeline) kind=${arg:-float64}; td=/tmp/ssa_ft_${tag}_${kind}; rm -rf $td
              here=$PWD; (cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --output-format csv -d $td -- python3 $here/tools/fact_timeline.py run $kind) > $out/timeline_$kind.log 2>&1
              echo "timeline[$kind] trace rc=$?" >> $sum
              nf=2; unk=""; [ "$kind" = stack4 ] && { nf=4; unk="24571 24571 24571 24571"; }; [ "$kind" = single ] && { nf=1; unk="41419"; }
              timeout 600 python tools/fact_timeline.py rounds $td $nf > $out/round_timeline_$kind.txt 2>&1
              timeout 600 python tools/fact_timeline.py head $td $nf > $out/head_timeline_$kind.txt 2>&1
              [ "$kind" != float32 ] && timeout 900 python tools/fact_timeline.py table $td $unk > $out/syrk_launch_table_$kind.txt 2>&1
              echo "timeline[$kind] $(tail -2 $out/syrk_launch_table_$kind.txt 2>/dev/null | head -1)" >> $sum ;;
    passes)   td=/tmp/ssa_pt_${tag}; rm -rf $td; here=$PWD
              (cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --output-format csv -d $td -- python3 $here/tools/pass_trace.py) > $out/pass_trace_run.log 2>&1
              timeout 300 python tools/pass_trace.py analyse $td > $out/pass_trace.txt 2>&1
              echo "passes rc=$? $(head -1 $out/pass_trace.txt)" >> $sum ;;
    ab)       log=$out/ab_$(echo "$arg" | tr ' /="' '____' | cut -c1-60).log
              eval "timeout 1500 python -X faulthandler tools/ab_knobs.py $arg" > $log 2>&1
              echo "ab[$arg] rc=$?" >> $sum; cat $log >> $sum ;;
    py)       log=$out/$(echo "$arg" | tr ' /' '__').log
              timeout 1200 python -X faulthandler $arg > $log 2>&1
              echo "py[$arg] rc=$? $(tail -1 $log)" >> $sum ;;
    *)        echo "unknown step $step" >> $sum ;;
  esac
done
cat $sum
