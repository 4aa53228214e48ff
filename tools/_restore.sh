# sourced by the A/B scripts that rebuild matten_amd/libmatten_hip.so in place (groups_ab.sh, ab5.sh, ab_train.sh, tp_trace.sh):
# whatever ends the script -- its last line, an error, Ctrl-C, the `timeout` these scripts run under -- the committed cg_gen.h is
# put back, the variant objects are removed and the production library is rebuilt.  (ops.tp_fused also refuses a library whose
# coupling code was generated for other groups than plan.TP_GROUPS: matten_tp_groups_hash.)
#   usage, from matten_amd/csrc:   source ../../tools/_restore.sh
_AB_SAVED_CG=$(mktemp /tmp/cg_gen_saved.XXXXXX.h)
cp cg_gen.h "$_AB_SAVED_CG"
_ab_restore() {
  trap - EXIT INT TERM HUP
  cmp -s "$_AB_SAVED_CG" cg_gen.h || cp "$_AB_SAVED_CG" cg_gen.h
  rm -f "$_AB_SAVED_CG" build/*_[0-9].o build/*_[0-9][0-9].o build/*_[A-E].o
  # AB_NO_RESTORE=1: skip the rebuild (an ephemeral gpurun box is thrown away after the call; the committed cg_gen.h is still put back)
  [ -n "$AB_NO_RESTORE" ] || make -B -j8 > /dev/null 2>&1
}
trap _ab_restore EXIT
trap 'exit 130' INT TERM HUP
