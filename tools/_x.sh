for v in "" x_noadd x_norot x_noperm x_all ""; do
  if [ -n "$v" ]; then export TSGU_LIB_PATH=$PWD/build/variants/$v.so; else unset TSGU_LIB_PATH; fi
  echo "== variant '$v'"; python tools/linemarch_check.py --time 2>&1 | grep "^linemarch.*forward"
done
