/*
 * TEST INFRASTRUCTURE ONLY (oracle/Makefile.mi355x) -- the table edits of integration/mi355x-tables.patch without touching the
 * reference tree.  /root/reference is read-only and its sources may not be copied, so the build that boots the reference
 * runtime on the GPU factories cannot apply that patch to a copy.  Instead this header is force-included (gcc -include) in
 * front of ONE reference file, machine-description/ocr-machine-description.c -- the only user of the four name tables and
 * factory switches (its create_factory_guid / _workpile / _schedulerHeuristic / _commplatform, :320-530, :872) -- and
 * redirects, for that translation unit only:
 *     <family>_types[]        -> <family>_types_mi355x[]        the stock names followed by the GPU one (tables.c fills it from
 *                                                               the stock table at load time: no name list is restated here)
 *     <family>Max_id          -> <family>Max_id + 1             so that TO_ENUM (:91-95) walks one entry further
 *     new<Family>Factory()    -> new<Family>FactoryMi355x()     the GPU id goes to integration/'s constructor, any other id to the
 *                                                               stock switch
 * which is, entry for entry, what the patch does inside guid-all.{h,c}, scheduler-heuristic-all.{h,c}, workpile-all.{h,c} and
 * comm-platform-all.{h,c}.  The GPU ids are the old <family>Max_id values, exactly the values the patch's enum entries get.
 */
#ifndef OCR_MI355X_TABLES_H
#define OCR_MI355X_TABLES_H

#include "guid/guid-all.h"
#include "scheduler-heuristic/scheduler-heuristic-all.h"
#include "workpile/workpile-all.h"
#include "comm-platform/comm-platform-all.h"

enum { guidGpuHash_id = guidMax_id, schedulerHeuristicGpu_id = schedulerHeuristicMax_id, workpileGpu_id = workpileMax_id,
       commPlatformRccl_id = commPlatformMax_id };

extern const char *guid_types_mi355x[], *schedulerHeuristic_types_mi355x[], *workpile_types_mi355x[], *commplatform_types_mi355x[];

ocrGuidProviderFactory_t *newGuidProviderFactoryMi355x(guidType_t type, ocrParamList_t *typeArg);
ocrSchedulerHeuristicFactory_t *newSchedulerHeuristicFactoryMi355x(schedulerHeuristicType_t type, ocrParamList_t *perType);
ocrWorkpileFactory_t *newWorkpileFactoryMi355x(workpileType_t type, ocrParamList_t *perType);
ocrCommPlatformFactory_t *newCommPlatformFactoryMi355x(commPlatformType_t type, ocrParamList_t *typeArg);

/* the constructors of integration/ (what the patch declares in the <family>-all.h headers) */
ocrGuidProviderFactory_t *newGuidProviderFactoryGpuHash(ocrParamList_t *typeArg, u32 factoryId);
ocrSchedulerHeuristicFactory_t *newOcrSchedulerHeuristicFactoryGpu(ocrParamList_t *perType, u32 factoryId);
ocrWorkpileFactory_t *newOcrWorkpileFactoryGpu(ocrParamList_t *perType);
ocrCommPlatformFactory_t *newCommPlatformFactoryRccl(ocrParamList_t *perType);

#ifndef OCR_MI355X_TABLES_IMPL   /* (a macro's own name is not expanded again inside it: the enumerators stay what they are) */
#define guidMax_id               ((guidType_t)(guidMax_id + 1))
#define schedulerHeuristicMax_id ((schedulerHeuristicType_t)(schedulerHeuristicMax_id + 1))
#define workpileMax_id           ((workpileType_t)(workpileMax_id + 1))
#define commPlatformMax_id       ((commPlatformType_t)(commPlatformMax_id + 1))
#define guid_types               guid_types_mi355x
#define schedulerHeuristic_types schedulerHeuristic_types_mi355x
#define workpile_types           workpile_types_mi355x
#define commplatform_types       commplatform_types_mi355x
#define newGuidProviderFactory        newGuidProviderFactoryMi355x
#define newSchedulerHeuristicFactory  newSchedulerHeuristicFactoryMi355x
#define newWorkpileFactory            newWorkpileFactoryMi355x
#define newCommPlatformFactory        newCommPlatformFactoryMi355x
#endif
#endif
