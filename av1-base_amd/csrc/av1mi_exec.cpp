// av1mi_exec.cpp - av1mi_job_execute(): the encode segment of the reference's JobExecutor::execute
// (/root/reference/crates/daemon/src/job_executor.rs:266-317, 413-436) around the in-process encoder, in the
// reference's order and with its stage names and failure strings (job_executor.rs:71-81).  What follows the
// segment in the reference (size gate :319-327, atomic replacement :336-340, skip markers :389-399) is the daemon's
// control plane and stays with the caller (SURVEY.md §8 scope).
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <dirent.h>

#include <chrono>
#include <string>

#include "../../include/av1mi.h"

namespace {

struct ExecState {
  av1mi_state_cb cb;
  void *user;
  av1mi_job_metrics m;
  std::chrono::steady_clock::time_point t0;
  double fps_n_over_d;
};

void set_stage(ExecState &st, const char *stage) {
  snprintf(st.m.stage, sizeof(st.m.stage), "%s", stage);
  if (st.cb) st.cb(st.user, st.m.stage, &st.m);
}

// progress callback of av1mi_encode_file -> JobMetrics.{progress, fps, frames_encoded, total_frames, bitrate_kbps}
void on_progress(void *user, uint32_t done, uint32_t total, double fps, uint64_t bytes) {
  ExecState &st = *(ExecState *)user;
  st.m.frames_encoded = done;
  st.m.total_frames = total;
  st.m.progress = total ? (float)done / (float)total : 0.f;
  st.m.fps = (float)fps;
  st.m.bitrate_kbps = done ? (float)((double)bytes * 8.0 / 1000.0 / ((double)done / st.fps_n_over_d)) : 0.f;  // at the clip's own frame rate
  if (st.cb) st.cb(st.user, st.m.stage, &st.m);
}

// std::fs::remove_dir_all
void remove_dir_all(const std::string &dir) {
  DIR *d = opendir(dir.c_str());
  if (!d) return;
  while (dirent *e = readdir(d)) {
    if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
    const std::string p = dir + "/" + e->d_name;
    struct stat sb;
    if (lstat(p.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode)) remove_dir_all(p);
    else unlink(p.c_str());
  }
  closedir(d);
  rmdir(dir.c_str());
}

int mkdir_p(const std::string &dir) {
  for (size_t i = 1; i <= dir.size(); i++) {
    if (i == dir.size() || dir[i] == '/') {
      const std::string sub = dir.substr(0, i);
      if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return -errno;
    }
  }
  return 0;
}

}  // namespace

extern "C" int av1mi_job_execute(const av1mi_exec_job *job, av1mi_state_cb state_cb, void *user, av1mi_job_metrics *metrics,
                                 char *error, size_t error_cap) {
  if (error && error_cap) error[0] = 0;
  if (!job || !job->id || !job->input_path || !job->output_path || !job->temp_base_dir) return AV1MI_E_INVALID_ARG;
  ExecState st = {};
  st.cb = state_cb; st.user = user;
  st.m.crf = job->params.cq_level; st.m.workers = job->workers;
  auto fail = [&](int code, const std::string &msg) {
    if (error && error_cap) snprintf(error, error_cap, "%s", msg.c_str());
    set_stage(st, "failed");                         // JobState::Failed(msg)
    if (metrics) *metrics = st.m;
    return code;
  };
  {  // the clip's frame rate (bitrate) - the reference has no source for it (JobMetrics.bitrate_kbps stays 0, job_executor.rs:117-137); here from the Y4M header
    av1mi_clip_info ci = {};
    st.fps_n_over_d = (av1mi_probe_y4m(job->input_path, &ci) == AV1MI_OK && ci.fps_num && ci.fps_den) ? (double)ci.fps_num / ci.fps_den : 30.0;
    st.m.total_frames = ci.frames;
  }
  set_stage(st, "encoding");                          // job.state = JobState::Encoding (job_executor.rs:270)
  const std::string chunks = std::string(job->temp_base_dir) + "/chunks_" + job->id;  // :274
  int rc = mkdir_p(chunks);                           // create_dir_all -> JobError::TempDirCreation
  if (rc) return fail(rc, std::string("IO error: ") + strerror(-rc));
  av1mi_job ej = {};
  ej.input_path = job->input_path; ej.output_path = job->output_path; ej.temp_dir = chunks.c_str();
  ej.workers = job->workers; ej.chunk_frames = 0; ej.gpu_mask = 0; ej.params = job->params;
  av1mi_report rep = {};
  rc = av1mi_encode_file(&ej, on_progress, &st, &rep);  // spawn_blocking(run_av1an(&params)) (:287)
  if (rc != 0) {                                      // Ok(Err(encode_err)) (:413-423): state Failed(err), temp dir removed
    remove_dir_all(chunks);
    if (rc < 0) return fail(rc, std::string("IO error: ") + strerror(-rc));                      // EncodeError::Io
    char msg[96];
    snprintf(msg, sizeof(msg), "MI355X encoder failed with exit code: %d", rc);                   // EncodeError::Av1anFailed(code)
    return fail(rc, msg);
  }
  st.m.psnr = (float)rep.psnr[0];
  set_stage(st, "validating");                        // :291
  struct stat sb;
  if (stat(job->output_path, &sb) != 0) {             // :296-306
    const std::string msg = std::string("Output file not found: ") + strerror(errno);
    remove_dir_all(chunks);
    return fail(AV1MI_E_FORMAT, msg);
  }
  if (sb.st_size == 0) {                              // :308-317
    remove_dir_all(chunks);
    unlink(job->output_path);
    return fail(AV1MI_E_FORMAT, "Output file is empty");
  }
  st.m.size_in_bytes_after = (uint64_t)sb.st_size;
  st.m.progress = 1.f;
  remove_dir_all(chunks);                             // the reference removes it after replacement (:351); nothing of ours is left in it
  set_stage(st, "size_gating");                       // hands back to the caller at :319
  if (metrics) *metrics = st.m;
  return AV1MI_OK;
}
