/* written by ace-compiler_amd/build.py */
static const char fpr_[] = "ACEHIP_SRC_FPR=a1cafe80fa0623d4";
const char* acehip_source_fingerprint(void) { return fpr_ + 15; }
