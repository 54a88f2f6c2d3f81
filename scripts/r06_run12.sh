echo "== rows of 16 slots"; python scripts/micro/turbo_perm_probe.py 2>&1 | grep -v shuffle | tail -4
echo "== rows of GW slots"; CPX_TURBO_RW=gw python scripts/micro/turbo_perm_probe.py 2>&1 | grep -v shuffle | tail -4
CPX_TURBO_RW=gw python scripts/micro/turbo_slot_probe.py 2>&1 | tail -3
