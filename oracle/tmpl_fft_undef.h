/* oracle/tmpl_fft_undef.h -- TEST INFRASTRUCTURE ONLY: undefines every macro of a tmpl_fft.h instantiation. */
#undef TNAME
#undef T_FE
#undef S_FE
#undef S_ONE
#undef T_ZERO
#undef T_ADD
#undef T_SUB
#undef S_MUL
#undef T_MULS
#undef S_INV
#undef S_POW64
#undef S_ROOT_OF_UNITY
#undef S_GENERATOR
#undef S_FROM_U64
#undef S_S
#undef FFT_UNDEF_ALL
