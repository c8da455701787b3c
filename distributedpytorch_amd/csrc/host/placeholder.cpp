extern "C" int dpa_host_abi_version() { return 1; }
