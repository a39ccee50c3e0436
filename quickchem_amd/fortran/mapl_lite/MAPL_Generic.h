!  mapl_lite/MAPL_Generic.h -- the error-handling macros the grid components are written with, under the names MAPL's
!  own MAPL_Generic.h gives them (OH_GridCompMod.F90:1 includes that header; :514,528 and every call site use
!  __Iam__, __RC__, __STAT__, VERIFY_, _ASSERT, RETURN_).  Inside GEOS the real header is found first; here these
!  definitions serve, on top of MAPL_VRFY / MAPL_ASRT / MAPL_RTRN of this directory's module MAPL.  As in MAPL's
!  ErrLog.h the traceback names the FILE and line, not Iam: QC_Environment/QC_EnvironmentMod.F90:44 uses __RC__ in a
!  procedure that declares no Iam (and takes STATUS from its module), and must compile against this header unmodified.
!     __Iam__('name')   declares STATUS and the traceback name Iam
!     f(..., __RC__)    passes RC=STATUS and leaves the caller, with STATUS as its RC, when it is not zero
!     allocate(x, __STAT__)   the same for STAT=
!     VERIFY_(status)   leaves the caller when status is not zero
!     _ASSERT(cond, 'message')   prints the message and leaves the caller with a failure code when cond is false
!     RETURN_(code)     sets the (optional) RC and returns
#ifndef MAPL_LITE_GENERIC_H
#define MAPL_LITE_GENERIC_H
#define __Iam__(name) integer :: STATUS; character(len=255) :: Iam=name
#define VERIFY_(A) if(MAPL_VRFY(A,__FILE__,__LINE__,RC))return
#define _ASSERT(A,msg) if(MAPL_ASRT(A,msg,__FILE__,__LINE__,RC))return
#define RETURN_(A) if(MAPL_RTRN(A,__FILE__,__LINE__,RC))return
#define __RC__ RC=STATUS); VERIFY_(STATUS
#define __STAT__ STAT=STATUS); VERIFY_(STATUS
#endif
