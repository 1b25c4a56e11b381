#ifndef RUTILS_STUB_H
#define RUTILS_STUB_H
void R_CheckUserInterrupt(void);
#endif
