// limg_hip_rccl.h -- RCCL entry points resolved at run time.  liblimg_hip.so has no link-time dependency on librccl: a host program that already carries
// an RCCL (PyTorch-ROCm bundles its own copy) must not end up with two, so the symbols come from whatever `librccl.so.1` the process has or can load
// (the loader hands back an already-loaded object with that SONAME).  Types from <rccl/rccl.h>; nothing here is part of the C ABI.
#ifndef LIMG_HIP_RCCL_H
#define LIMG_HIP_RCCL_H

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>

namespace limg_hip
{
  struct Rccl
  {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;      // (the three below are evidence for the bench line, not needed by the data path)
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok = false;
  };

  inline const Rccl &rccl()
  {
    static Rccl r = [] {
      Rccl x;
      void *h = nullptr;
      for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
      if (!h)
      {
        fprintf(stderr, "limg_hip: RCCL not found (%s)\n", dlerror());
        return x;
      }
#define LIMG_RCCL_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, sym))
      LIMG_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); LIMG_RCCL_SYM(CommInitRank, "ncclCommInitRank"); LIMG_RCCL_SYM(CommDestroy, "ncclCommDestroy");
      LIMG_RCCL_SYM(AllGather, "ncclAllGather"); LIMG_RCCL_SYM(Send, "ncclSend"); LIMG_RCCL_SYM(Recv, "ncclRecv");
      LIMG_RCCL_SYM(GroupStart, "ncclGroupStart"); LIMG_RCCL_SYM(GroupEnd, "ncclGroupEnd"); LIMG_RCCL_SYM(GetErrorString, "ncclGetErrorString");
      LIMG_RCCL_SYM(CommCount, "ncclCommCount"); LIMG_RCCL_SYM(CommUserRank, "ncclCommUserRank"); LIMG_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef LIMG_RCCL_SYM
      x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.Send && x.Recv && x.GroupStart && x.GroupEnd && x.GetErrorString;
      return x;
    }();
    return r;
  }
}

#endif
