// Logger -- same convention as the reference's src/logger.hpp:8-73: a mutex-protected singleton,
// LOG_DEBUG/INFO/WARN/ERROR macros, and LOG_ERROR latching HasError()/GetLastError().  Unlike the
// reference the per-frame INFO lines can be silenced (SetMinLevel): the reference's ~9 LOG_INFO lines
// per frame (src/scaler.cpp:262,360,465-477) take a mutex and format a timestamp each.
#pragma once
#include <chrono>
#include <ctime>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>

class Logger {
public:
    enum class Level { DEBUG, INFO, WARNING, ERROR };

    static Logger& Get() {
        static Logger instance;
        return instance;
    }

    template <typename... Args>
    void Log(Level level, Args&&... args) {
        if (level < m_minLevel && level != Level::ERROR) return;
        std::lock_guard<std::mutex> lock(m_mutex);
        std::stringstream ss;
        ss << "[" << GetTimestamp() << "] " << GetLevelString(level) << ": ";
        (ss << ... << std::forward<Args>(args));
        std::cout << ss.str() << std::endl;
        if (level == Level::ERROR) {
            m_hasError = true;
            m_lastError = ss.str();
        }
    }

    bool HasError() const { return m_hasError; }
    std::string GetLastError() const { return m_lastError; }
    void ClearError() { m_hasError = false; m_lastError.clear(); }
    void SetMinLevel(Level level) { m_minLevel = level; }

private:
    Logger() = default;

    std::string GetTimestamp() {
        auto now = std::chrono::system_clock::now();
        auto time = std::chrono::system_clock::to_time_t(now);
        char buffer[26];
        ctime_r(&time, buffer);
        buffer[24] = '\0';
        return buffer;
    }

    const char* GetLevelString(Level level) {
        switch (level) {
            case Level::DEBUG: return "DEBUG";
            case Level::INFO: return "INFO";
            case Level::WARNING: return "WARNING";
            case Level::ERROR: return "ERROR";
            default: return "UNKNOWN";
        }
    }

    std::mutex m_mutex;
    bool m_hasError = false;
    std::string m_lastError;
    Level m_minLevel = Level::INFO;
};

#define LOG_DEBUG(...) Logger::Get().Log(Logger::Level::DEBUG, __VA_ARGS__)
#define LOG_INFO(...) Logger::Get().Log(Logger::Level::INFO, __VA_ARGS__)
#define LOG_WARN(...) Logger::Get().Log(Logger::Level::WARNING, __VA_ARGS__)
#define LOG_ERROR(...) Logger::Get().Log(Logger::Level::ERROR, __VA_ARGS__)
