#ifndef RINTERNALS_STUB_H
#define RINTERNALS_STUB_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NamesSymbol;
int TYPEOF(SEXP);
R_xlen_t XLENGTH(SEXP);
double* REAL(SEXP);
int* INTEGER(SEXP);
SEXP Rf_install(const char*);
int R_has_slot(SEXP, SEXP);
SEXP R_do_slot(SEXP, SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP Rf_mkChar(const char*);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
Rboolean Rf_isMatrix(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asLogical(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
Rboolean R_ToplevelExec(void (*)(void*), void*);
void Rf_onintr(void);
#endif
