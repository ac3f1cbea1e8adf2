P=tools/queue_probe.py
for cfg in "HP_HALO_VARIANT=0 HP_HALO_KG=1" "HP_HALO_VARIANT=1 HP_HALO_KG=1" "HP_HALO_VARIANT=0 HP_HALO_KG=2" "HP_HALO_VARIANT=1 HP_HALO_KG=2" "HP_HALO_VARIANT=1 HP_HALO_KG=2 HP_HALO_BK=128" "HP_HALO_VARIANT=0 HP_HALO_KG=2 HP_HALO_BK=128"; do
  env $cfg python $P --pipes 4 --modes injected,engine
done
