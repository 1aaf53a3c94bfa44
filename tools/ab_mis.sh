for L in prev new prev new; do
  if [ $L = prev ]; then export AQ_LIB=$PWD/atlasqtl_amd/libatlasqtl_hip_prev.so; else unset AQ_LIB; fi
  echo $L $(timeout -k 10 300 python tools/dev_check.py mis 2>&1 | grep -o "core [0-9.]* ms" | tr '\n' ' ')
done
