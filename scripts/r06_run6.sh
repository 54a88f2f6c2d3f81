for v in v_ld64 v_st64 v_fb; do echo "== $v"; CPX_LIB_PATH=$PWD/ab/$v.so python scripts/micro/turbo_slot_probe.py 2>&1 | tail -3; done
